#!/bin/bash
# Round 5: where does the patterns input's 9 % between round 3 and round 4 come from -- the kernel / layout, or the VI loop around
# it?  The sparse pass alone (tools/probe/pass_standalone.py) and bench.py's fit, both under rocprofv3 --kernel-trace --stats, for
# round 3's tree (_ab/r03), the commit before and at the step (_ab/c_3682606, _ab/c_1982841) and HEAD.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_standalone; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stat() {  # tag
  python3 - $OUT/$1 <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'loglik_stream_kernel' in r['Name'] or 'fwd_apply' in r['Name'] or 'update_k' in r['Name']:
            print('   %-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
for t in r03:_ab/r03 c36:_ab/c_3682606 c19:_ab/c_1982841 head:.; do
  name=${t%%:*}; dir=$R/${t##*:}
  cd $dir
  echo "== $name: the pass alone"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone_$name -- python3 $R/tools/probe/pass_standalone.py > /dev/null 2>&1
  stat alone_$name
  echo "== $name: inside the fit (bench.py, patterns)"
  ARGS="--workload c2 --steps 60 --warmup 5 --cpu-steps 0 --prewarm 100"
  [ $name = head ] && ARGS="$ARGS --generator patterns --no-by-input"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_$name -- python3 bench.py $ARGS > /dev/null 2>&1
  stat fit_$name
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
