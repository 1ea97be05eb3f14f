set -e
OUT=gpurun_out/r02p
mkdir -p $OUT
timeout -k 10 500 python bench.py --steps 1 --warmup 1 --no-bc --no-cpu --cpu-curve '' --force-slab > $OUT/bench_slab.json 2> $OUT/bench_slab.err || { tail -20 $OUT/bench_slab.err; exit 1; }
python -c "
import json; d=json.load(open('$OUT/bench_slab.json')); print(d['value'], d.get('slab_rag'))"
