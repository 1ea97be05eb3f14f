set -e
OUT=gpurun_out/r02m
mkdir -p $OUT
GLIA_HMT_LIB=$GRAFT_REPO_ROOT/glia_amd/libglia_hmt_r1.so timeout -k 10 200 python tools/bc_bench.py 512 16 > $OUT/bc512_r1lib.txt 2>&1 || true
grep -v amdgpu $OUT/bc512_r1lib.txt | tail -3 | cut -c1-200
timeout -k 10 200 python tools/bc_bench.py 512 16 > $OUT/bc512_now.txt 2>&1
grep -v amdgpu $OUT/bc512_now.txt | tail -2 | cut -c1-200
