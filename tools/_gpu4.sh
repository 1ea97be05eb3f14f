cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout -k 10 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_relabel.py -x -q -m gpu > gpurun_out/r02d/pytest.log 2>&1; tail -4 gpurun_out/r02d/pytest.log
for sz in 256 512 1024; do
  GLIA_PB_HASH=1 timeout -k 10 300 python tools/pb_bench.py $sz 16 2 > gpurun_out/r02d/pb_window_$sz.txt 2>&1
  tail -2 gpurun_out/r02d/pb_window_$sz.txt
done
GLIA_HMT_LIB=$PWD/glia_amd/libglia_hmt_prof.so timeout -k 10 300 python tools/pb_bench.py 1024 16 2 > gpurun_out/r02d/prof_window_1024.txt 2>&1
grep -E "window profile|merges/s" gpurun_out/r02d/prof_window_1024.txt | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc1 -o pmc1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pb_bench.py 1024 16 2 > $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc2 -o pmc2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pb_bench.py 1024 16 2 > $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc2.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc1 $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc2
