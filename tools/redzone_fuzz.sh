# light fuzz with junk behind the end of the merge loops' device blocks: bash tools/redzone_fuzz.sh <hexmask> <seconds> <seed>
mkdir -p gpurun_out
GLIA_HMT_REDZONE=1 GLIA_HMT_REDZONE_MASK=$1 FUZZ_LIGHT=1 timeout -k 10 $(( $2 + 100 )) python tests/fuzz_gpu.py $2 $3 > gpurun_out/fuzz_redzone_$1.txt 2>&1
grep -v "^Exception\|^TypeError\|^Traceback\|^  File" gpurun_out/fuzz_redzone_$1.txt | tail -n 1 | cut -c1-300
