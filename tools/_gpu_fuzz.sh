set -e
mkdir -p gpurun_out/r02k
timeout -k 10 700 python tests/fuzz_gpu.py 600 20261005 > gpurun_out/r02k/fuzz.txt 2>&1
tail -3 gpurun_out/r02k/fuzz.txt
