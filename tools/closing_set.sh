set -e
mkdir -p gpurun_out/r01k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r01k/bench1024.json 2> gpurun_out/r01k/bench1024.err
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01k/prof -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r01k/prof_bench.log 2>&1
echo stats done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01k/pmc_fetch -o acc -- python3 tools/acc_bench.py 1024 16 1 > gpurun_out/r01k/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01k/pmc_write -o acc -- python3 tools/acc_bench.py 1024 16 1 > gpurun_out/r01k/pmc_write.log 2>&1
echo pmc done
timeout -k 10 200 python tools/bc_bench.py 512 16 > gpurun_out/r01k/bc512_final.txt 2>&1
timeout -k 10 200 python tools/bc_bench.py 256 16 > gpurun_out/r01k/bc256_final.txt 2>&1
timeout -k 10 300 python tools/pb_bench.py 512 16 1 > gpurun_out/r01k/pbmed512_final.txt 2>&1
echo bc done
