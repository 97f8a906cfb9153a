#!/bin/bash
# diagnostic build with in-kernel cycle stamps, run, and rebuild the product library (GPU box)
set -e
cd $GRAFT_REPO_ROOT/polee_amd/csrc
touch loglik.hip && make -s -j8 EXTRA=-DPOLEE_STAMPS > /dev/null
cd $GRAFT_REPO_ROOT
POLEE_DEBUG_PRINT=1 python3 tools/probe/stamps.py "$@"
cd polee_amd/csrc && touch loglik.hip && make -s -j8 > /dev/null
