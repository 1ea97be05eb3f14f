set -e
for v in expG; do
  GLIA_HMT_LIB=$GRAFT_REPO_ROOT/glia_amd/libglia_hmt_$v.so timeout -k 10 200 python tools/bc_bench.py 512 16 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150 | awk -v v=$v '{print v, $0}'
done
timeout -k 10 200 python tools/bc_bench.py 512 16 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150 | awk '{print "now", $0}'
