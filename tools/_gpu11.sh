set -e
OUT=gpurun_out/r02l
mkdir -p $OUT
GLIA_HMT_LIBM=device timeout -k 10 200 python tools/bc_bench.py 512 16 > $OUT/bc512_devlibm.txt 2>&1
grep -v amdgpu $OUT/bc512_devlibm.txt | cut -c1-200
git_rev=none
