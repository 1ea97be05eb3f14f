# The GPU tests of the loop kernels against the wave-skew builds (glia_amd/csrc/skew.hpp; make -C glia_amd/csrc skew skew2), once each,
# through gpurun:  bash tools/skew_tests.sh <tag>
# skew  = behind every workgroup barrier every wave but wave 0 sleeps ~32 k cycles (the thread that rewrites a shared word is early),
# skew2 = wave 0 alone sleeps (it is late).  The tests compare with the oracle / between queues byte for byte, as always; the session
# ends with the check that no call returned GLIA_HMT_ERR_INTERNAL (tests/conftest.py).
set -e
TAG=${1:-skew}
mkdir -p gpurun_out/$TAG
for v in skew skew2; do
  GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_$v.so timeout -k 10 560 python -m pytest tests/test_gpu_queue.py tests/test_gpu_merge.py tests/test_gpu_golden.py tests/test_gpu_bc.py tests/test_gpu_watershed.py tests/test_gpu_rag.py \
    -m gpu -x -q -k "not config2 and not 128_cubed" --durations=5 > gpurun_out/$TAG/pytest_$v.txt 2>&1 || { tail -30 gpurun_out/$TAG/pytest_$v.txt; exit 1; }
  echo "$v: $(tail -1 gpurun_out/$TAG/pytest_$v.txt)"
done
