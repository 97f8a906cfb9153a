#!/bin/bash
# Round 4: packing-policy sweep (environment knobs of psell_build.cpp, no rebuild) over the inputs of the diversity table.
# usage: tools/probe/r04_sweep.sh <tag> "<VAR=val VAR=val>|<...>" [inputs...]
cd $GRAFT_REPO_ROOT
TAG=$1; SETS=$2; shift; shift
INPUTS=${*:-"p0 literal fixture p0.3"}
OUT=gpurun_out/sweep_$TAG; mkdir -p $OUT
IFS='|' read -ra ARR <<< "$SETS"
for input in $INPUTS; do
  case $input in
    p0) ARGS="--workload c2 --generator patterns --no-by-input" ;;
    p0.1) ARGS="--workload c2 --generator patterns --set-diversity 0.1" ;;
    p0.3) ARGS="--workload c2 --generator patterns --set-diversity 0.3" ;;
    literal) ARGS="--workload c2 --generator literal --no-by-input" ;;
    fixture) ARGS="--workload fixture" ;;
  esac
  for i in "${!ARR[@]}"; do
    set_=${ARR[$i]}
    env $set_ timeout 600 python3 bench.py $ARGS --steps ${STEPS:-50} --warmup 5 --cpu-steps 0 --prewarm ${PREWARM:-300} 2> $OUT/${input}_$i.err | tail -1 > $OUT/${input}_$i.json
    python3 - "$input" "$set_" $OUT/${input}_$i.json <<'PY'
import sys, json
try:
    j = json.loads(open(sys.argv[3]).read()); r = j['roofline']
    print('%-8s %-60s it/s %5.0f kernel %.4f pass %.4f phys GB %.3f frac %.3f B/nnz %.2f shares %s' % (sys.argv[1], sys.argv[2] or '(default)', j['value'], r['kernel_ms_avg'], r['pass_ms_avg'],
          r['physical_bytes_per_launch'] / 1e9, r['frac'], r['layout_bytes_per_nnz'], [round(v, 3) for v in r['stream_share_of_nnz'][:5]]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'failed', e)
PY
  done
done | tee $OUT/summary.txt
