#!/bin/bash
# Round 4, first GPU call: the GPU test-suite with stream S (collapsed single-transcript fragments), the set-diversity
# sweep, and a clock read per tile (pre-built diagnostic library) on the inputs the generator does not flatter.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04a/gputests.log 2>&1; echo "gpu tests rc $?" >> gpurun_out/r04a/gputests.log
tail -3 gpurun_out/r04a/gputests.log
timeout 1500 bash tools/probe/diversity_bench.sh r04a c2 > gpurun_out/r04a/diversity.md 2>&1
cat gpurun_out/r04a/diversity.md
for mode in literal fixture 0.3 0; do
  POLEE_HIP_LIB=$GRAFT_REPO_ROOT/polee_amd/csrc/libpolee_hip_tilecycles.so timeout 600 python3 tools/probe/tile_cycles.py $mode > gpurun_out/r04a/tile_cycles_$mode.txt 2>&1
  cat gpurun_out/r04a/tile_cycles_$mode.txt | tail -8
done
# packing-policy knobs on the literal input: dense unions instead of masked slices
for g in 0.3 0.5 10; do
  echo "MASK_GAIN=$g" ; POLEE_PSELL_MASK_GAIN=$g timeout 600 python3 bench.py --literal-subsets --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('it/s %.0f kernel ms %.4f physical GB/launch %.3f frac %.3f shares %s' % (j['value'], r['kernel_ms_avg'], r['physical_bytes_per_launch'] / 1e9, r['frac'], [round(v, 3) for v in r['stream_share_of_nnz']]))"
done > gpurun_out/r04a/mask_gain_literal.txt 2>&1
cat gpurun_out/r04a/mask_gain_literal.txt
