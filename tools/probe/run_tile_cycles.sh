#!/bin/bash
# diagnostic build with in-kernel cycle counters, per-tile statistics, then the product library again (GPU box)
set -e
cd $GRAFT_REPO_ROOT/polee_amd/csrc
touch loglik.hip && make -s -j8 EXTRA=-DPOLEE_TILE_CYCLES > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/probe/tile_cycles.py "$@" > gpurun_out/tile_cycles_${TAG:-x}.txt 2>&1 || true
cd polee_amd/csrc && touch loglik.hip && make -s -j8 > /dev/null 2>&1
