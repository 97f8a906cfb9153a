#!/bin/bash
# Round 5: same-box A/B of LIBRARIES (POLEE_HIP_LIB) and builder / kernel environment knobs over several inputs, the
# configurations run round-robin REPS times so that clock drift hits all of them alike.
# usage: tools/probe/r05_ab.sh <tag> "<name;lib.so;VAR=val VAR=val>|<...>" [inputs...]      (lib relative to polee_amd/csrc)
cd $GRAFT_REPO_ROOT
TAG=$1; SETS=$2; shift; shift
INPUTS=${*:-"literal p0 wide fixture"}
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
IFS='|' read -ra ARR <<< "$SETS"
for input in $INPUTS; do
  case $input in
    p0) ARGS="--workload c2 --generator patterns --no-by-input" ;;
    p0.3) ARGS="--workload c2 --generator patterns --set-diversity 0.3" ;;
    literal) ARGS="--workload c2 --generator literal --no-by-input" ;;
    wide) ARGS="--workload wide --generator literal --no-by-input" ;;
    fixture) ARGS="--workload fixture" ;;
    fixture_full) ARGS="--workload fixture_full" ;;
    c5) ARGS="--workload c5 --generator literal --no-by-input" ;;
  esac
  for rep in $(seq 1 ${REPS:-2}); do
    for i in "${!ARR[@]}"; do
      IFS=';' read -r name lib envs <<< "${ARR[$i]}"
      env POLEE_HIP_LIB=$GRAFT_REPO_ROOT/polee_amd/csrc/$lib $envs timeout -k 5 200 python3 bench.py $ARGS $EXTRA_ARGS --steps ${STEPS:-100} --warmup 5 --cpu-steps 0 --prewarm ${PREWARM:-300} 2> $OUT/${input}_${i}_$rep.err | tail -1 > $OUT/${input}_${i}_$rep.json
      python3 - "$input" "$name" $OUT/${input}_${i}_$rep.json <<'PY'
import sys, json
try:
    j = json.loads(open(sys.argv[3]).read()); r = j['roofline']
    print('%-8s %-28s it/s %6.0f kernel %.4f pass %.4f step %.4f phys GB %.3f frac %.3f eff %.3f tiles %d shares %s' % (sys.argv[1], sys.argv[2], j['value'], r['kernel_ms_avg'], r['pass_ms_avg'], j['ms_per_step'],
          r['physical_bytes_per_launch'] / 1e9, r['frac'], r['effective_frac'], j['detail']['num_tiles'], [round(v, 3) for v in r['stream_share_of_nnz'][:5]]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'failed', e)
PY
    done
  done
done | tee $OUT/summary.txt
