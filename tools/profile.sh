#!/bin/bash
# Profiles bench.py's hot kernel with rocprofv3 on the GPU box; writes CSV summaries under gpurun_out/prof_<tag>/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5  # (scratch of earlier runs would mix into the summary)
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-steps 0 --prewarm 100 --no-by-input $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc2.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc5 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc5.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc3.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc4 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc4.err
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
# kernel stats
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    out['kernel_stats'] = rows[:15]
# per-kernel MEDIAN duration from the dispatch trace (the stats CSV only has the mean, which one slow first launch shifts)
for f in glob.glob('trace/**/*kernel_trace.csv', recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        try:
            dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError):
            pass
    med = {}
    for k, v in dur.items():
        v.sort()
        med[k] = {'calls': len(v), 'median_ns': v[len(v) // 2], 'mean_ns': sum(v) / len(v), 'min_ns': v[0], 'max_ns': v[-1]}
    out['kernel_medians'] = med
for d in ('pmc1','pmc2','pmc3','pmc4','pmc5'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            if 'loglik' in k or 'scan' in k or 'vi_' in k:
                out.setdefault('pmc', {}).setdefault(k, {}).update({c: sum(v)/len(v) for c, v in cs.items()})
json.dump(out, open('summary.json','w'), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
# keep only small files
find $OUT -name "*.csv" -size +2M -delete
