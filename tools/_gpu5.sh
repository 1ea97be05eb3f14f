set -e
OUT=gpurun_out/r02e
mkdir -p $OUT
GLIA_HMT_LIB=$GRAFT_REPO_ROOT/glia_amd/libglia_hmt_prof.so timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_prof.txt 2>&1
grep "wide phases\|scan phases" $OUT/pb1024_prof.txt | tail -2
