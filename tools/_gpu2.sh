cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout -k 10 600 python -m pytest tests/test_gpu_merge.py -x -q -m gpu > gpurun_out/r02b/pytest_merge.log 2>&1; tail -3 gpurun_out/r02b/pytest_merge.log
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 300 python tools/pb_bench.py 1024 16 2 > gpurun_out/r02b/prof_window_1024.txt 2>&1
grep -E "profile|merges/s" gpurun_out/r02b/prof_window_1024.txt | tail -6
GLIA_HMT_PB_WINDOW=0 GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 300 python tools/pb_bench.py 1024 16 2 > gpurun_out/r02b/prof_tree_1024.txt 2>&1
grep -E "profile|merges/s" gpurun_out/r02b/prof_tree_1024.txt | tail -12
