# the whole -m gpu suite in one process, output under gpurun_out/<tag>/ (run through gpurun): bash tools/gpu_tests.sh <tag>
set -e
TAG=${1:-tests}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$TAG/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/$TAG/pytest_gpu.txt
