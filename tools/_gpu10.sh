set -e
OUT=gpurun_out/r02k
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_bc.py tests/test_libm.py tests/test_gpu_golden.py tests/test_gpu_cli.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
GLIA_HMT_BC_NOCOMMON=1 timeout -k 10 600 python -m pytest tests/test_gpu_bc.py -x -q -m gpu > $OUT/pytest_nocommon.txt 2>&1 || { tail -30 $OUT/pytest_nocommon.txt; exit 1; }
tail -1 $OUT/pytest_nocommon.txt
GLIA_HMT_BC_GENERIC=1 timeout -k 10 600 python -m pytest tests/test_gpu_bc.py -x -q -m gpu > $OUT/pytest_generic.txt 2>&1 || { tail -30 $OUT/pytest_generic.txt; exit 1; }
tail -1 $OUT/pytest_generic.txt
timeout -k 10 200 python tools/bc_bench.py 512 16 > $OUT/bc512.txt 2>&1
grep -v amdgpu $OUT/bc512.txt | cut -c1-220
