#!/bin/bash
# Round 5: is the streaming kernel (92 KB of code) bound by instruction fetch?  Lists the SQC / instruction-cache counters this
# rocprofv3 knows and collects them for the sparse pass on one input.   usage: r05_icache.sh [bench args]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_icache; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQC_|INST_FETCH|WAIT_IFETCH|SQ_WAIT_INST|INST_LEVEL" | head -80 > $OUT/counters.txt
ARGS="--steps 6 --warmup 2 --cpu-steps 0 --prewarm 50 --no-by-input $*"
i=0
for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/err$i.txt
  python3 - $OUT/p$i <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        if 'stream' in k:
            for c, v in sorted(cs.items()): print('%-34s %16.0f' % (c, sum(v)/len(v)))
PY
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
