cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout -k 10 900 python -m pytest tests/test_gpu_bc.py tests/test_gpu_cli.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r02o/pytest.log 2>&1; tail -12 gpurun_out/r02o/pytest.log
