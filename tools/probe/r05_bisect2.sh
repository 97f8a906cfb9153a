#!/bin/bash
# Round 5, second step of the same-box bisect: the gene-patterns input on round 3's last commit, four commits of round 4 that
# touched the streaming kernel (worktrees under _ab/c_<commit>, each with its own library and bench.py), round 4's library and HEAD.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_bisect2; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
run() {  # name dir lib "args" rep
  ( cd $2 && env POLEE_HIP_LIB=$3 timeout 900 python3 bench.py $4 --steps ${STEPS:-100} --warmup 5 --cpu-steps 0 --prewarm 300 2> $R/$OUT/$1_$5.err | tail -1 > $R/$OUT/$1_$5.json )
  python3 - "$1" $OUT/$1_$5.json <<'PY'
import sys, json
try:
    j = json.loads(open(sys.argv[2]).read()); r = j['roofline']
    print('patterns %-10s it/s %6.0f kernel %.4f pass %.4f step %.4f phys GB %.3f tiles %s' % (sys.argv[1], j['value'], r['kernel_ms_avg'], r['pass_ms_avg'], j['ms_per_step'],
          r.get('physical_bytes_per_launch', 0) / 1e9, j['detail'].get('num_tiles')))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
for rep in $(seq 1 ${REPS:-2}); do
  run r03 $R/_ab/r03 $R/_ab/r03/polee_amd/csrc/libpolee_hip.so "--workload c2" $rep
  for c in 3682606 1982841; do run c_$c $R/_ab/c_$c $R/_ab/c_$c/polee_amd/csrc/libpolee_hip.so "--workload c2" $rep; done
  for c in 88c5671 50b362e; do run c_$c $R/_ab/c_$c $R/_ab/c_$c/polee_amd/csrc/libpolee_hip.so "--workload c2 --generator patterns --no-by-input" $rep; done
  run r04 $R $R/polee_amd/csrc/libpolee_hip_r04.so "--workload c2 --generator patterns --no-by-input" $rep
  run head $R $R/polee_amd/csrc/libpolee_hip.so "--workload c2 --generator patterns --no-by-input" $rep
done | tee $OUT/summary.txt
