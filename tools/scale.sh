#!/bin/bash
# First contact with a multi-GPU node: ONE command that runs 1 / 2 / 4 / 8 ranks x {C2 one sample per GPU
# (weak, no collective), C2 one sample row-sharded (strong: one all-reduce of K n f32 per pass), C3 (6 samples sharded, strong),
# C4 (8 samples per GPU, weak)} and collects the bench lines -- each carrying config.comm = what the transport itself reports
# (polee_comm_info: RCCL's ncclCommCount / ncclCommUserRank) -- into gpurun_out/scale/<workload>_<ranks>.json + summary.txt.
#   usage: tools/scale.sh [max ranks, default: the GPUs visible]       env: STEPS (default 100), RANKS ("1 2 4 8"),
#          POLEE_COMM_ALGO=rs_ag (the gradient's exchange as reduce-scatter + all-gather instead of one all-reduce)
# Nothing here computes an efficiency: the driver does that from the per-N values.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
OUT=${OUT:-$ROOT/gpurun_out/scale}; mkdir -p "$OUT"
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
MAX=${1:-$NGPU}
WORKLOADS=${WORKLOADS:-"weak rowshard c3 c4"}
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > "$OUT/summary.txt"
for N in ${RANKS:-1 2 4 8}; do
  [ "$N" -gt "$MAX" ] && continue
  for W in $WORKLOADS; do
    case $W in
      weak) ARGS="--workload c2 --no-by-input --cpu-steps 0" ;;
      rowshard) ARGS="--workload c2 --row-shard --no-by-input --cpu-steps 0" ;;
      c3) ARGS="--workload c3" ;;
      c4) ARGS="--workload c4" ;;
    esac
    [ "$W" = rowshard ] && [ "$N" = 1 ] && continue
    timeout ${TIMEOUT:-1200} python3 bench.py --gpus $N --steps ${STEPS:-100} --warmup 5 $ARGS 2> "$OUT/${W}_$N.err" | tail -1 > "$OUT/${W}_$N.json"
    python3 - "$W" "$N" "$OUT/${W}_$N.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    j = json.loads(open(sys.argv[3]).read())
    if j.get("dry_run"):  # (POLEE_BENCH_DRY=1: the launch path only -- tests/test_multiproc.py)
        print("%-9s ranks %s: dry run, %d ranks seen" % (sys.argv[1], sys.argv[2], j["ranks_seen"]))
    else:
        print("%-9s ranks %s: %10.1f %s  (%.4f ms/step, scaling %s, comm %s)" % (sys.argv[1], sys.argv[2], j["value"], j["unit"], j["ms_per_step"], j["scaling"], j["config"].get("comm")))
except Exception as e:
    print("%-9s ranks %s: FAILED (%s)" % (sys.argv[1], sys.argv[2], e))
PY
  done
done
