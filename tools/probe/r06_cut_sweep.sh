#!/bin/bash
# the waves' shares of a tile (wave_cuts, loglik.hip): cost model a + b x groups-of-four + c x (1 + groups) at a run's start
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_cut_sweep.txt; : > $OUT
cd $GRAFT_REPO_ROOT
for cfg in "" "POLEE_BYTES_CUT=1" "POLEE_CUT_MODEL=10,6,0" "POLEE_CUT_MODEL=10,6,3" "POLEE_CUT_MODEL=10,6,6" "POLEE_CUT_MODEL=20,6,6" "POLEE_CUT_MODEL=6,6,10" "POLEE_CUT_MODEL=30,6,3" "" "$@"; do
  for gen in literal patterns; do
    line=$(env $cfg timeout 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-by-input --generator $gen 2>/dev/null | tail -1)
    echo "$cfg $gen" $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f  kernel %.4f ms' % (d['value'], d['roofline']['kernel_ms_avg']))") | tee -a $OUT
  done
done
