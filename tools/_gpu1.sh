set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bc.py::test_fuzz_near_tie_case_of_round_1 > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
for sz in 256 512 1024; do
  GLIA_PB_HASH=1 GLIA_HMT_PB_WINDOW=0 timeout -k 10 300 python tools/pb_bench.py $sz 16 2 > gpurun_out/r02a/pb_tree_$sz.txt 2>&1
  GLIA_PB_HASH=1 timeout -k 10 300 python tools/pb_bench.py $sz 16 2 > gpurun_out/r02a/pb_window_$sz.txt 2>&1
  tail -2 gpurun_out/r02a/pb_tree_$sz.txt; tail -2 gpurun_out/r02a/pb_window_$sz.txt
done
