#!/bin/bash
# Round 5: same-box comparison of round 3's last commit (61c84f0, a worktree under _ab/r03 with its own library,
# bench.py and generator), round 4's library (libpolee_hip_r04.so) and HEAD on three inputs, alternating, REPS times each.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_bisect; mkdir -p $OUT
run() {  # name dir lib "args" input rep
  ( cd $2 && env POLEE_HIP_LIB=$3 timeout 900 python3 bench.py $4 --steps ${STEPS:-100} --warmup 5 --cpu-steps 0 --prewarm 300 2> $GRAFT_REPO_ROOT/$OUT/$5_$1_$6.err | tail -1 > $GRAFT_REPO_ROOT/$OUT/$5_$1_$6.json )
  python3 - "$5" "$1" $OUT/$5_$1_$6.json <<'PY'
import sys, json
try:
    j = json.loads(open(sys.argv[3]).read()); r = j['roofline']
    print('%-8s %-6s it/s %6.0f kernel %.4f pass %.4f step %.4f phys GB %.3f tiles %s' % (sys.argv[1], sys.argv[2], j['value'], r['kernel_ms_avg'], r['pass_ms_avg'], j['ms_per_step'],
          r.get('physical_bytes_per_launch', 0) / 1e9, j['detail'].get('num_tiles')))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'failed', e)
PY
}
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-5}); do
  for input in patterns literal fixture; do
    case $input in
      patterns) A3="--workload c2"; A4="--workload c2 --generator patterns --no-by-input" ;;
      literal) A3="--workload c2 --literal-subsets"; A4="--workload c2 --generator literal --no-by-input" ;;
      fixture) A3="--workload fixture"; A4="--workload fixture" ;;
    esac
    run r03 $R/_ab/r03 $R/_ab/r03/polee_amd/csrc/libpolee_hip.so "$A3" $input $rep
    run r04 $R $R/polee_amd/csrc/libpolee_hip_r04.so "$A4" $input $rep
    run head $R $R/polee_amd/csrc/libpolee_hip.so "$A4" $input $rep
  done
done | tee $OUT/summary.txt
