#!/bin/bash
# Round 4: GPU test-suite with the leaf-order fit, then A/B of the leaf-order mode (same box), then per-kernel times.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gpu tests rc $?" >> $OUT/gputests.log
tail -5 $OUT/gputests.log
bash tools/probe/r04_sweep.sh leaf "|POLEE_VI_NO_LEAF_ORDER=1||POLEE_VI_NO_LEAF_ORDER=1" p0 fixture literal
cd /tmp && export TMPDIR=/tmp
for w in c2 fixture; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 50 --warmup 5 --cpu-steps 0 --prewarm 100 > $GRAFT_REPO_ROOT/$OUT/bench_trace_$w.json 2> $GRAFT_REPO_ROOT/$OUT/trace_$w.err
  f=$(find $GRAFT_REPO_ROOT/$OUT/trace_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$OUT/kernel_stats_$w.csv && head -14 $f | cut -c1-200
  find $GRAFT_REPO_ROOT/$OUT/trace_$w -name "*.csv" -size +1M -delete
done
