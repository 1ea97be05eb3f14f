# accumulation-pass experiments (through gpurun): bash tools/acc_variants.sh <tag> [variant names ...]
# tools/acc_bench.py 1024 16 3 on the shipped library and on every glia_amd/libglia_hmt_acc_<name>.so (make -C glia_amd/csrc accvar NAME=.. ACCFLAGS=..)
set -e
TAG=${1:-acc}; shift || true
mkdir -p gpurun_out/$TAG
if [ -z "$NO_TESTS" ]; then timeout -k 10 300 python -u -m pytest tests/test_gpu_rag.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/$TAG/pytest_rag.txt; fi
timeout -k 10 120 python tools/acc_bench.py 1024 16 3 2>&1 | tail -1 | sed "s/^/default: /" | tee gpurun_out/$TAG/acc.txt
for v in "$@"; do
  GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_acc_$v.so timeout -k 10 120 python tools/acc_bench.py 1024 16 3 2>&1 | tail -1 | sed "s/^/$v: /" | tee -a gpurun_out/$TAG/acc.txt
done
