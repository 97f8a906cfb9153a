#!/bin/bash
# round 6's evidence in one GPU call: rocprofv3 trace + PMC passes of the default bench command, then the unprofiled bench lines
O=$GRAFT_REPO_ROOT/gpurun_out/r06_final
mkdir -p $O
cd $GRAFT_REPO_ROOT
tools/profile.sh r06 > $O/profile.log 2>&1
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 2> $O/bench_c2.err | tail -1 > $O/bench_c2.json
python3 bench.py --steps 500 --warmup 5 --cpu-steps 0 --no-by-input 2>/dev/null | tail -1 > $O/bench_c2_500.json
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-by-input --deterministic 2>/dev/null | tail -1 > $O/bench_c2_det.json
python3 bench.py --workload c5 --steps 20 --warmup 3 --cpu-steps 0 --no-by-input 2>/dev/null | tail -1 > $O/bench_c5.json
python3 bench.py --workload c3 --steps 300 --warmup 10 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --steps 50 --warmup 5 --cpu-steps 0 --no-by-input --samples-per-gpu 2 2>/dev/null | tail -1 > $O/bench_c2_cohort2.json
python3 tools/probe/create_time.py > $O/create_time.txt 2>&1
POLEE_PREP_TREE=cluster_auto python3 tools/probe/prep_throughput.py 16 4 > $O/cohort.txt 2>&1
python3 tools/probe/regression_bench.py 64 100 > $O/reg_s64.txt 2>&1
for f in bench_c2 bench_c2_500 bench_c2_det bench_c5 bench_c3 bench_c2_cohort2; do python3 -c "
import json,sys
d=json.load(open('$O/$f.json')); r=d.get('roofline',{})
print('$f', round(d['value'],1), d['unit'], 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f' % r.get('kernel_ms_avg',0), 'frac %.3f / %.3f' % (r.get('frac',0), r.get('effective_frac',0)))"; done | tee $O/lines.txt
tail -3 $O/create_time.txt; tail -3 $O/cohort.txt; cat $O/reg_s64.txt
