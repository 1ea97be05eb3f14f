set -e
mkdir -p gpurun_out/r02h
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02h/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/r02h/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/r02h/pytest_gpu.txt
