# quick gate for classifier-loop experiments (on the GPU box, via gpurun): bc tests + digests at 256^3 / 512^3.   bash tools/bc_check.sh <tag> [notests]
TAG=${1:-bc}
mkdir -p gpurun_out/$TAG
if [ "$2" != "notests" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_bc.py -x -q > gpurun_out/$TAG/pytest_bc.txt 2>&1; tail -3 gpurun_out/$TAG/pytest_bc.txt
fi
GLIA_BC_HASH=1 timeout -k 10 120 python tools/bc_bench.py 256 16 > gpurun_out/$TAG/bc256.txt 2>&1; tail -2 gpurun_out/$TAG/bc256.txt
GLIA_BC_HASH=1 timeout -k 10 200 python tools/bc_bench.py 512 16 > gpurun_out/$TAG/bc512.txt 2>&1; tail -2 gpurun_out/$TAG/bc512.txt
echo "expect 256: 51b7b5316e0d8fd648ab2b444527633d8eaf65c5 5eadbd683d6a042f7bf950aa2352ef93bdc12956"
echo "expect 512: 8e620b69eab2cc31991eaca662446f52ad2231df 8e30e7ba92d7b89dc09c413260a0841df1cbf886"
