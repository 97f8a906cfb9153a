#!/bin/bash
# the round's evidence in one GPU call: rocprofv3 trace + PMC passes, then the unprofiled bench lines
R=${1:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
cd $GRAFT_REPO_ROOT
tools/profile.sh $R > $O/profile.log 2>&1
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 2> $O/bench_c2.err | tail -1 > $O/bench_c2.json
python3 bench.py --steps 500 --warmup 5 --cpu-steps 0 2>/dev/null | tail -1 > $O/bench_c2_500.json
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --deterministic 2>/dev/null | tail -1 > $O/bench_c2_det.json
python3 bench.py --workload c5 --steps 20 --warmup 3 --cpu-steps 0 2>/dev/null | tail -1 > $O/bench_c5.json
python3 bench.py --workload c3 --steps 300 --warmup 10 2>/dev/null | tail -1 > $O/bench_c3.json
for s in 2 4; do python3 bench.py --steps 50 --warmup 5 --cpu-steps 0 --samples-per-gpu $s 2>/dev/null | tail -1 > $O/bench_c2_cohort$s.json; done
# the per-tile kernel alone over every slice (the cross-check algorithm; also what the mixed stream B runs on)
POLEE_NO_RING=1 python3 bench.py --steps 10 --warmup 2 --cpu-steps 0 --prewarm 20 2>/dev/null | tail -1 > $O/bench_c2_per_tile_kernel.json
# set diversity (generator as built, dropout 0.1 / 0.3, literal subsets, tiled real fixture): one table
tools/probe/diversity_bench.sh $R c2 > $O/diversity.md 2> $O/diversity.err
# X construction at C2 scale
python3 tools/probe/xbuild_bench.py 200000 30000000 > $O/xbuild.txt 2>&1
