#!/bin/bash
# the round's evidence in one GPU call: rocprofv3 trace + PMC passes, then the unprofiled bench lines
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
cd $GRAFT_REPO_ROOT
tools/profile.sh r02 > $O/profile.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 2> $O/bench_c2.err | tail -1 > $O/bench_c2.json
python bench.py --steps 500 --warmup 5 --cpu-steps 0 2>/dev/null | tail -1 > $O/bench_c2_500.json
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --deterministic 2>/dev/null | tail -1 > $O/bench_c2_det.json
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --samples-per-gpu 2 2>/dev/null | tail -1 > $O/bench_c2_cohort2.json
python bench.py --workload c5 --steps 20 --warmup 3 --cpu-steps 0 2>/dev/null | tail -1 > $O/bench_c5.json
python bench.py --workload c3 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
tools/probe/run_stamps.sh 2>&1 | grep -v -E "warning|NSTAMP|\^" | tail -18 > $O/stamps.txt
for s in 3 4; do python bench.py --steps 50 --warmup 5 --cpu-steps 0 --samples-per-gpu $s 2>/dev/null | tail -1 > $O/bench_c2_cohort$s.json; done
