# SQ counters of the accumulation pass at 1024^3 (run through gpurun): bash tools/acc_ablate.sh <tag>
set -e
TAG=${1:-acc}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/acc_bench.py 1024 16 3 2>&1 | tail -1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_sq -o acc -- python3 tools/acc_bench.py 1024 16 1 > gpurun_out/$TAG/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_sq2 -o acc -- python3 tools/acc_bench.py 1024 16 1 > gpurun_out/$TAG/pmc_sq2.log 2>&1 || true
python3 - <<PY
import csv,collections
for d in ('pmc_sq','pmc_sq2'):
    try: rows=list(csv.DictReader(open('gpurun_out/$TAG/%s/acc_counter_collection.csv'%d)))
    except Exception as e: print(d,'missing',e); continue
    acc=collections.defaultdict(list)
    for r in rows:
        if 'rag_accumulate' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v), 'per 64 voxels: %.1f' % (sum(v)/len(v)/16777216.0))
PY
