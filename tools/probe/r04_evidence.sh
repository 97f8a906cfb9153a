#!/bin/bash
# Round 4 evidence in one GPU call: the driver's bench configuration (headline + by_input), a 500-step fit, the profile
# (rocprofv3 kernel trace + separate PMC passes) of the headline input, the diversity table, C5, deterministic mode.
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -1 $OUT/bench_c2.json | cut -c1-600
python3 bench.py --steps 500 --warmup 5 --cpu-steps 0 --no-by-input > $OUT/bench_c2_500.json 2> $OUT/bench_c2_500.err
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-by-input --deterministic > $OUT/bench_c2_det.json 2> $OUT/bench_c2_det.err
python3 bench.py --workload c5 --steps 20 --warmup 5 --cpu-steps 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python3 bench.py --workload c5 --generator patterns --steps 20 --warmup 5 --cpu-steps 0 > $OUT/bench_c5_patterns.json 2> $OUT/bench_c5_patterns.err
python3 bench.py --samples-per-gpu 2 --steps 50 --warmup 5 --cpu-steps 0 > $OUT/bench_c2_cohort2.json 2> $OUT/bench_c2_cohort2.err
python3 bench.py --workload c3 --steps 300 --warmup 5 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
bash tools/probe/diversity_bench.sh $TAG c2 > $OUT/diversity.md 2>&1
cat $OUT/diversity.md
for f in $OUT/bench_*.json; do echo $f; tail -1 $f | python3 -c "
import sys, json
try:
    j = json.loads(sys.stdin.read()); r = j.get('roofline', {})
    print('  value %.1f %s  ms/step %.4f  kernel %.4f  frac %.3f  eff %.3f' % (j['value'], j['unit'], j['ms_per_step'], r.get('kernel_ms_avg', 0), r.get('frac', 0), r.get('effective_frac', 0)))
    for k, v in r.get('by_input', {}).items(): print('    ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ('value', 'kernel_ms_avg', 'pass_ms_avg', 'frac', 'effective_frac', 'error')})
    if 'cpu_baseline' in j: print('    cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
except Exception as e: print('  failed', e)
"; done
