#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "gpu tests rc $?" >> $OUT/gputests.log
tail -5 $OUT/gputests.log
OLD="POLEE_PSELL_SPLIT_MASKED=1 POLEE_PSELL_NO_INTERLEAVE=1"
bash tools/probe/r04_sweep.sh merge "|$OLD|POLEE_PSELL_NO_INTERLEAVE=1||$OLD" literal p0 fixture p0.3
