#!/bin/bash
# register / scratch usage of the streaming kernels (cross-compiles loglik.hip with resource remarks)
cd "$(dirname "$0")/../../polee_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $EXTRA -c loglik.hip -o /tmp/loglik_r.o -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
python3 - "$@" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else r"stream2?_kernelILi[46]E"
txt = open('/tmp/res.txt').read().split('remark: Function Name: ')[1:]
for blk in txt:
    name = blk.split()[0]
    if not re.search(pat, name): continue
    g = lambda k: re.search(k + r": (\d+)", blk)
    print("%-75s VGPR %s AGPR %s SGPR %s scratch %s occ %s LDS %s" % (name[:75], g(" VGPRs").group(1), g("AGPRs").group(1), g("SGPRs").group(1), g("ScratchSize \[bytes/lane\]").group(1), g("Occupancy \[waves/SIMD\]").group(1), g("LDS Size \[bytes/block\]").group(1)))
PY
