#!/bin/bash
# kernel table of the regression step at S = 64 (C4's per-GPU share x 8)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_reg64; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/probe/regression_bench.py 64 40 > $OUT/run.txt 2> $OUT/trace.err
cd $OUT; python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_trace.csv', recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        try: dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError): pass
    tab = sorted(((sum(v), k, len(v), sorted(v)[len(v)//2]) for k, v in dur.items()), reverse=True)
    with open('kernel_medians.txt', 'w') as o:
        for tot, k, n, med in tab[:24]:
            line = f'{k[:80]:80s} calls {n:5d} median {med/1e3:8.1f} us total {tot/1e6:8.2f} ms'
            print(line); o.write(line + '\n')
PY
find $OUT -name "*.csv" -size +2M -delete
