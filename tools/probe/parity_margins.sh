#!/bin/bash
# runs tools/probe/parity_margins.py on the GPU box; the report goes to gpurun_out/ (copied into profiles/ by hand)
cd $GRAFT_REPO_ROOT
python3 tools/probe/parity_margins.py "$@" 2>&1 | tee gpurun_out/r06_parity_margins.txt
echo "exit code ${PIPESTATUS[0]}" >> gpurun_out/r06_parity_margins.txt
