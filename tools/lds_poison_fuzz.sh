# light fuzz with the merge kernels' LDS poisoned at entry: bash tools/lds_poison_fuzz.sh <mode 1|2|3> <seconds> <seed>
mkdir -p gpurun_out
GLIA_HMT_LDS_POISON=$1 FUZZ_LIGHT=1 timeout -k 10 $(( $2 + 100 )) python tests/fuzz_gpu.py $2 $3 > gpurun_out/fuzz_ldspoison_$1.txt 2>&1
grep -v "^Exception\|^TypeError\|^Traceback\|^  File" gpurun_out/fuzz_ldspoison_$1.txt | tail -n 1 | cut -c1-300
