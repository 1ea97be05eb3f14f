cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02t
for r in 1500000 1000000 600000 300000; do GLIA_HMT_REBASE=$r timeout -k 10 300 python tools/pb_bench.py 1024 16 2 > gpurun_out/r02t/pb_1024_r$r.txt 2>&1; echo $r; tail -1 gpurun_out/r02t/pb_1024_r$r.txt | cut -c1-200; done
