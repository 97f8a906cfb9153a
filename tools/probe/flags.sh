#!/bin/bash
# compiler-flag sweep for one device file (FILE, default loglik.hip) on the GPU box: rebuilds the library per flag set
# (FLAGSETS="a|b|...", added to that file's flags only) and times the C2 fit
cd $GRAFT_REPO_ROOT
FILE=${FILE:-loglik.hip}
IFS="|" read -ra SETS <<< "${FLAGSETS:-}"; for fl in "${SETS[@]}"; do
  touch polee_amd/csrc/$FILE; rm -f polee_amd/csrc/_obj/${FILE%.hip}.o
  if make -s -C polee_amd/csrc _obj/${FILE%.hip}.o EXTRA="$fl" > /tmp/mk.log 2>&1 && make -s -C polee_amd/csrc >> /tmp/mk.log 2>&1; then
    echo "FLAGS [$fl]: $(timeout 600 python tools/probe/time_fit.py 2>&1 | tail -1)"
  else
    echo "FLAGS [$fl]: build failed: $(grep -m1 -i error /tmp/mk.log | cut -c1-120)"
  fi
done
touch polee_amd/csrc/$FILE; make -s -C polee_amd/csrc > /dev/null 2>&1
