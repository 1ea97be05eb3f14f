set -e
mkdir -p gpurun_out/a2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/acc_bench.py 1024 16 3 2>&1 | tail -1
for d in 0 1 2 3 8 11; do GLIA_HMT_DEBUG=$d GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 120 python tools/acc_bench.py 1024 16 2 2>&1 | tail -1; done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/a2/pmc_sq -o acc -- python3 tools/acc_bench.py 1024 16 1 > gpurun_out/a2/pmc_sq.log 2>&1
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/a2/pmc_sq/acc_counter_collection.csv')))
acc=collections.defaultdict(list)
for r in rows:
    if 'rag_accumulate' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
