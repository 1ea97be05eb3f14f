# the GPU measurements a round's profiles/ are built from, in two gpurun calls:
#   bash tools/closing_set.sh <tag> a     bench line, rocprofv3 kernel stats of the same command, the three PMC passes on the accumulation pass
#   bash tools/closing_set.sh <tag> b     loop digests and timings, phase counters (profiling build), config 5 end to end, watershed, transform
set -e
TAG=${1:-r04z}; PART=${2:-a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$PART" = a ]; then
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $OUT/bench1024.json 2> $OUT/bench1024.err
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-bc > $OUT/prof_bench.log 2>&1
echo stats done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o acc -- python3 tools/acc_bench.py 1024 16 1 > $OUT/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o acc -- python3 tools/acc_bench.py 1024 16 1 > $OUT/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -o acc -- python3 tools/acc_bench.py 1024 16 1 > $OUT/pmc_sq.log 2>&1
echo pmc done
else
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_hash.txt 2>&1
GLIA_PB_HASH=1 timeout -k 10 200 python tools/pb_bench.py 512 16 2 > $OUT/pb512_hash.txt 2>&1
GLIA_BC_HASH=1 timeout -k 10 200 python tools/bc_bench.py 1024 16 > $OUT/bc1024.txt 2>&1
GLIA_BC_HASH=1 timeout -k 10 200 python tools/bc_bench.py 512 16 > $OUT/bc512.txt 2>&1
GLIA_BC_HASH=1 timeout -k 10 200 python tools/bc_bench.py 256 16 > $OUT/bc256.txt 2>&1
echo loops done
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 200 python tools/bc_bench.py 1024 16 > $OUT/bc1024_prof.txt 2>&1 || true
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 200 python tools/pb_bench.py 1024 16 2 > $OUT/pb1024_prof.txt 2>&1 || true
echo counters done
timeout -k 10 300 python tools/e2e_bench.py > $OUT/e2e_2048x2048x512.txt 2>&1
timeout -k 10 300 python tools/ws_bench.py 256 512 1024 > $OUT/watershed.txt 2>&1
timeout -k 10 200 python tools/transform_bench.py > $OUT/transform.txt 2>&1
echo e2e done
fi
