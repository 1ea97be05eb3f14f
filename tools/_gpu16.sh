set -e
OUT=gpurun_out/r02q
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_queue.py tests/test_gpu_bc.py -x -q -m gpu -k "horizon or instances" > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
