# The GPU tests of the loop kernels against a wave-skew build (glia_amd/csrc/skew.hpp; make -C glia_amd/csrc skew skew2), through gpurun:
#   bash tools/skew_tests.sh <tag> skew|skew2
# skew  = behind every workgroup barrier every wave but wave 0 sleeps ~8 k cycles (the thread that rewrites a shared word is early),
# skew2 = wave 0 alone sleeps (it is late).  The tests compare with the oracle / between queues byte for byte, as always; the session
# ends with the check that no call returned GLIA_HMT_ERR_INTERNAL (tests/conftest.py).  -v + unbuffered: a line per test (the runner
# kills a command that stays silent for seven minutes).
set -e
TAG=${1:-skew}; V=${2:-skew}
mkdir -p gpurun_out/$TAG
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_$V.so PYTHONUNBUFFERED=1 timeout -k 10 1100 python -u -m pytest tests/test_gpu_queue.py tests/test_gpu_merge.py tests/test_gpu_golden.py tests/test_gpu_bc.py \
    -m gpu -x -v -k "not cli" --durations=8 2>&1 | tee gpurun_out/$TAG/pytest_$V.txt | grep -E "PASSED|FAILED|ERROR|passed|failed" | awk '{n++; if (n % 10 == 0 || /passed|failed|FAILED|ERROR/) print}'
