#!/bin/bash
# round-6 iteration loop on the GPU box: the VI parity tests, one bench line, one kernel trace of the same command.
# usage: tools/probe/r06_iter.sh <tag> [tests|notests] [extra bench args...]
set -u
TAG=${1:-it}; MODE=${2:-tests}; shift; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$MODE" = tests ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "vi or fit or optimize or degenerate or trajectory or replay or c1 or c2 or cohort or fast_" > $OUT/tests.log 2>&1
  echo "tests rc=$?" >> $OUT/tests.log; tail -5 $OUT/tests.log
fi
timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-by-input "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-steps 0 --prewarm 100 --no-by-input "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_trace.csv', recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        try: dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError): pass
    rows = []
    for k, v in dur.items():
        v.sort()
        rows.append((sum(v), k[:70], len(v), v[len(v)//2], sum(v)/len(v)))
    rows.sort(reverse=True)
    with open('kernel_medians.txt', 'w') as o:
        for tot, k, n, med, mean in rows[:25]:
            line = f'{k:70s} calls {n:5d} median {med/1e3:9.1f} us mean {mean/1e3:9.1f} us'
            print(line); o.write(line + '\n')
PY
find $OUT -name "*.csv" -size +2M -delete
