#!/bin/bash
# runs tools/probe/vi_stamps.py with the stamped library (built beforehand: hipcc ... -DPOLEE_VI_STAMPS -c vi.hip, linked with
# the product's other objects into polee_amd/csrc/libpolee_hip_vistamps.so)
cd $GRAFT_REPO_ROOT
POLEE_HIP_LIB=$GRAFT_REPO_ROOT/polee_amd/csrc/libpolee_hip_vistamps.so python3 tools/probe/vi_stamps.py 2>&1 | tee gpurun_out/r06_vi_stamps.txt
