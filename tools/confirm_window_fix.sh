# Confirms (or refutes) the fix of the window kernel's race: the poisoned light fuzz on the committed library and on the
# control build that has the barrier compiled out.  Run through gpurun, after `make -C glia_amd/csrc ctrl` in the container:
#   gpurun --timeout 1100 -- 'bash tools/confirm_window_fix.sh 450'
# Expected: the committed library clean, the control failing about once in 2 000 cases (a few failures in 450 s) -- every failure a pre_merge.
# GLIA_HMT_PREMERGE_ONCE=1 switches the second run of pre_merge off, or the net would hide what is being measured; the order replay cannot be
# switched off, so a failure of the control shows either as a MISMATCH or as a retry count above zero in the last line.
set -u
SECS=${1:-450}
mkdir -p gpurun_out
for which in fixed ctrl; do
  lib=$PWD/glia_amd/libglia_hmt.so; [ $which = ctrl ] && lib=$PWD/glia_amd/libglia_hmt_ctrl.so
  [ -f $lib ] || { echo "$lib missing (make -C glia_amd/csrc ctrl)"; continue; }
  GLIA_HMT_LIB=$lib GLIA_HMT_PREMERGE_ONCE=1 GLIA_HMT_POISON=rand FUZZ_LIGHT=1 timeout -k 10 $(( SECS + 100 )) python tests/fuzz_gpu.py $SECS 4242 > gpurun_out/confirm_$which.txt 2>&1
  echo "== $which: $(grep -v "^Exception\|^TypeError\|^Traceback\|^  File" gpurun_out/confirm_$which.txt | tail -n 1 | cut -c1-400)"
done
