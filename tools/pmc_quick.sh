#!/bin/bash
# quick PMC pass over bench.py: tools/pmc_quick.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-steps 0 "$@" > /dev/null 2> $OUT/err.txt
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('p/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        if 'stream' in k:
            for c, v in sorted(cs.items()): print('%-28s %16.0f' % (c, sum(v)/len(v)))
PY
find $OUT -name "*.csv" -size +2M -delete
