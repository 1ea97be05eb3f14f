set -e
OUT=gpurun_out/r02g
mkdir -p $OUT
GLIA_HMT_LIB=$GRAFT_REPO_ROOT/glia_amd/libglia_hmt_prof.so timeout -k 10 300 python tools/bc_bench.py 512 16 > $OUT/bc512_prof.txt 2>&1
grep "bc profile\|merges/s" $OUT/bc512_prof.txt | tail -8
