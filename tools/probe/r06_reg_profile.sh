#!/bin/bash
# kernel table of the regression step: bench.py --workload c3 (S = 6) and c4's per-GPU share at S = 64 (tools/probe/regression_bench.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_reg_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 300 --warmup 20 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 400 $OUT/bench_c3.json
python3 $GRAFT_REPO_ROOT/tools/probe/regression_bench.py 64 100 2>&1 | tee $OUT/s64.txt
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 100 --warmup 20 > /dev/null 2> $OUT/trace.err
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    dur = collections.defaultdict(list)
    for r in rows:
        try: dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError): pass
    tab = sorted(((sum(v), k, len(v), sorted(v)[len(v)//2]) for k, v in dur.items()), reverse=True)
    with open('kernel_medians.txt', 'w') as o:
        for tot, k, n, med in tab[:40]:
            line = f'{k[:90]:90s} calls {n:5d} median {med/1e3:8.1f} us total {tot/1e6:8.2f} ms'
            print(line); o.write(line + '\n')
PY
find $OUT -name "*.csv" -size +2M -delete
