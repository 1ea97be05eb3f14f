cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q
timeout -k 10 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_relabel.py -x -q -m gpu > gpurun_out/r02q/pytest.log 2>&1; tail -4 gpurun_out/r02q/pytest.log
for sz in 256 512 1024; do
  GLIA_PB_HASH=1 timeout -k 10 300 python tools/pb_bench.py $sz 16 2 > gpurun_out/r02q/pb_window_$sz.txt 2>&1
  tail -2 gpurun_out/r02q/pb_window_$sz.txt
done
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 300 python tools/pb_bench.py 1024 16 2 > gpurun_out/r02q/prof_window_1024.txt 2>&1
grep -E "batch profile|window profile|merges/s" gpurun_out/r02q/prof_window_1024.txt | tail -3
