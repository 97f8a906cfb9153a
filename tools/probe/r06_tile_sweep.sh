#!/bin/bash
# slices per tile (POLEE_TILE_A1 / A2 / A2M, honoured by both layout builders): C2 literal bench line per setting
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_tile_sweep.txt; : > $OUT
cd $GRAFT_REPO_ROOT
for cfg in "" "POLEE_TILE_A1=96" "POLEE_TILE_A1=128" "POLEE_TILE_A1=128 POLEE_TILE_A2=48" "POLEE_TILE_A1=96 POLEE_TILE_A2=48 POLEE_TILE_A2M=32" "POLEE_TILE_A1=48" "$@"; do
  line=$(env $cfg timeout 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-by-input 2>/dev/null | tail -1)
  echo "$cfg" $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f  step %.4f ms  kernel %.4f ms  tiles %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['detail']['num_tiles']))") | tee -a $OUT
done
