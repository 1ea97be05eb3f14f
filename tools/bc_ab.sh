# A/B of two libraries on the classifier loop (box-to-box and run-to-run noise is +-3 %: alternate, several times, compare minima)
#   bash tools/bc_ab.sh <libA.so> <libB.so> [size=512] [reps=3]
A=$1; B=$2; SIZE=${3:-512}; REPS=${4:-3}
for r in $(seq $REPS); do for v in $A $B; do
  GLIA_HMT_LIB=$v GLIA_BC_HASH=1 python -u tools/bc_bench.py $SIZE 16 2>&1 | awk -v v=$v '/^size/{l=$0} /^sha1/{h=$3} END{match(l,/loop [0-9.]+/); print v, substr(l,RSTART,RLENGTH), h}'
done; done
