# ablations of the accumulation pass on the profiling build (through gpurun): bash tools/acc_ablate2.sh <tag>
set -e
TAG=${1:-abl}
mkdir -p gpurun_out/$TAG
P=$PWD/glia_amd/libglia_hmt_prof.so
for d in 0 8 64 3 11 32; do
  GLIA_HMT_DEBUG=$d GLIA_HMT_LIB=$P timeout -k 10 120 python tools/acc_bench.py 1024 16 2 2>&1 | tail -4 | tee -a gpurun_out/$TAG/ablate.txt
done
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_acc_nodrain.so timeout -k 10 120 python tools/acc_bench.py 1024 16 2 2>&1 | tail -1 | sed 's/^/nodrain: /' | tee -a gpurun_out/$TAG/ablate.txt
