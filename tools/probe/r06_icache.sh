#!/bin/bash
# instruction-cache counters of the VI loop's tree kernels (they start behind a 92 KB streaming kernel: an I-cache of 64 KB per two CUs)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_icache; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/pmc
rocprofv3 --pmc SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_REQ SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-steps 0 --prewarm 50 --no-by-input > /dev/null 2> $OUT/err.txt
cd $OUT; python3 - <<'PY' | tee summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'vi_' in k or 'xwin' in k or 'loglik_stream' in k:
            acc[k.split('<')[0].split('::')[-1]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print('%-24s waves %6.0f  icache req %9.0f miss %7.0f dup %7.0f  ifetch_level %10.0f  wave_cycles %11.0f  (ifetch / wave cycles %.3f)' % (
        k, m.get('SQ_WAVES', 0), m.get('SQC_ICACHE_REQ', 0), m.get('SQC_ICACHE_MISSES', 0), m.get('SQC_ICACHE_MISSES_DUPLICATE', 0),
        m.get('SQ_IFETCH_LEVEL', 0), m.get('SQ_WAVE_CYCLES', 0), m.get('SQ_IFETCH_LEVEL', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1)))
PY
find $OUT -name "*.csv" -size +2M -delete
