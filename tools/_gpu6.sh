set -e
OUT=gpurun_out/r02f
mkdir -p $OUT
timeout -k 10 300 python tools/step_breakdown.py 1024 16 > $OUT/step_breakdown.txt 2>&1
grep -v amdgpu $OUT/step_breakdown.txt
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-bc --cpu-curve '' > $OUT/bench_nobc.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench_nobc.json')); print(d['value'], d['ms_per_step'], d['phases_ms'])"
