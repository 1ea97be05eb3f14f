cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
timeout -k 10 700 python tests/fuzz_gpu.py 600 20261004 > gpurun_out/r02p/fuzz_600s.txt 2>&1; tail -3 gpurun_out/r02p/fuzz_600s.txt
