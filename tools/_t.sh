set -e
mkdir -p gpurun_out/r02x
timeout -k 10 700 python tests/fuzz_gpu.py 600 20261006 > gpurun_out/r02x/fuzz.txt 2>&1
tail -2 gpurun_out/r02x/fuzz.txt
