#!/bin/bash
# A/B of builder / kernel switches (environment variables) on one box: tools/probe/exp_env.sh <tag> "<bench args>" "VAR=1 VAR2=x" "..." ...
TAG=$1; ARGS=$2; shift; shift
OUT=gpurun_out/exp_$TAG; mkdir -p $OUT
i=0
for ENVS in "$@"; do
  i=$((i+1))
  env $ENVS python3 bench.py --steps ${STEPS:-150} --warmup 5 --cpu-steps 0 --prewarm 200 $ARGS > $OUT/$i.json 2> $OUT/$i.err
  python3 - "$ENVS" $OUT/$i.json <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); r = j["roofline"]
    print("%-60s kernel %.4f pass %.4f it/s %.0f  B/nnz %.2f shares %s" % (sys.argv[1], r["kernel_ms_avg"], r["pass_ms_avg"], j["value"], r["layout_bytes_per_nnz"], " ".join("%.3f" % x for x in r["stream_share_of_nnz"])))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
