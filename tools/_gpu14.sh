set -e
OUT=gpurun_out/r02o
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_rag.py tests/test_gpu_golden.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
timeout -k 10 200 python tools/acc_bench.py 1024 16 5 2>&1 | grep -v amdgpu | tail -4
