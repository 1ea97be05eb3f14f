set -e
OUT=gpurun_out/r02f
mkdir -p $OUT
timeout -k 10 500 python bench.py --steps 5 --warmup 1 --no-bc --no-cpu --cpu-curve '' > $OUT/bench_nobc.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench_nobc.json')); print(d['value'], d['ms_per_step'], d['phases_ms'], d['host_call_ms'])"
