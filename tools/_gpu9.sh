set -e
OUT=gpurun_out/r02j
mkdir -p $OUT
for rb in 400000 800000 1600000 3200000; do
  for hz in 1 2 4; do
    GLIA_HMT_REBASE=$rb GLIA_HMT_HORIZON=$hz timeout -k 10 100 python tools/pb_bench.py 1024 16 2 2>/dev/null | tail -1 | awk -v rb=$rb -v hz=$hz '{print "rebase", rb, "horizon", hz, $0}' | cut -c1-170 >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
