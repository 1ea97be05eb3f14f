# light fuzz under GLIA_HMT_POISON=rand with a mask of poisoned blocks: bash tools/poison_bisect.sh <hexmask> <seconds> <seed>
mkdir -p gpurun_out
GLIA_HMT_POISON=rand GLIA_HMT_POISON_MASK=$1 FUZZ_LIGHT=1 timeout -k 10 $(( $2 + 100 )) python tests/fuzz_gpu.py $2 $3 > gpurun_out/fuzz_poison_$1.txt 2>&1
grep -v "^Exception\|^TypeError\|^Traceback\|^  File" gpurun_out/fuzz_poison_$1.txt | tail -1 | cut -c1-600
