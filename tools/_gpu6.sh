cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02r/pytest.log 2>&1; tail -4 gpurun_out/r02r/pytest.log
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/bench.err; tail -c 1500 gpurun_out/r02r/bench.json
