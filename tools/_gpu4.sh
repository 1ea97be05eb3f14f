set -e
OUT=gpurun_out/r02d
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_queue.py tests/test_gpu_merge.py tests/test_gpu_golden.py -x -q > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_hash.txt 2>&1
grep -v amdgpu $OUT/pb1024_hash.txt | cut -c1-250
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 512 16 2 > $OUT/pb512_hash.txt 2>&1
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 256 16 2 > $OUT/pb256_hash.txt 2>&1
GLIA_HMT_LIB=$GRAFT_REPO_ROOT/glia_amd/libglia_hmt_prof.so timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_prof.txt 2>&1
echo done
