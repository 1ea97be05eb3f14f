set -e
OUT=gpurun_out/r02i
mkdir -p $OUT
timeout -k 10 300 python -m cProfile -o $OUT/bench.prof bench.py --steps 4 --warmup 1 --no-bc --no-cpu --cpu-curve '' > $OUT/bench.json 2> $OUT/bench.err
python - <<'P' > $OUT/prof.txt
import pstats
p=pstats.Stats('gpurun_out/r02i/bench.prof'); p.sort_stats('cumulative').print_stats(35)
P
grep -v "^$" $OUT/prof.txt | head -60
