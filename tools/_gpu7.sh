cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
timeout -k 10 600 python -m pytest tests/test_gpu_rag.py -x -q -m gpu > gpurun_out/r02n/pytest.log 2>&1; tail -12 gpurun_out/r02n/pytest.log
timeout -k 10 400 python bench.py --steps 1 --warmup 0 --no-cpu --no-bc --force-slab --size 512 > gpurun_out/r02n/bench_slab.json 2> gpurun_out/r02n/bench_slab.err; tail -c 1200 gpurun_out/r02n/bench_slab.json; tail -5 gpurun_out/r02n/bench_slab.err
