set -e
OUT=gpurun_out/r02b
mkdir -p $OUT
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_hash.txt 2>&1
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 512 16 2 > $OUT/pb512_hash.txt 2>&1
GLIA_HMT_PB_BATCH=0 timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_seqwindow.txt 2>&1
GLIA_HMT_LIB=$GRAFT_REPO_ROOT/glia_amd/libglia_hmt_prof.so timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_prof.txt 2>&1
timeout -k 10 200 python tools/bc_bench.py 512 16 > $OUT/bc512.txt 2>&1
echo done
