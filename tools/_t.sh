set -e
OUT=gpurun_out/r02w
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_queue.py tests/test_gpu_merge.py tests/test_gpu_golden.py -x -q > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 1024 16 2 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 512 16 2 2>&1 | grep sha1 | tail -1
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 256 16 2 2>&1 | grep sha1 | tail -1
