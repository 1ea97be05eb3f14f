cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02s
timeout -k 10 900 python -m pytest tests/test_gpu_queue.py tests/test_gpu_merge.py -x -q -m gpu > gpurun_out/r02s/pytest.log 2>&1; tail -15 gpurun_out/r02s/pytest.log
