#!/bin/bash
# development shortcut: recompile vi.hip only (the one source that includes vi_fused.hpp) and relink both libraries
set -e
cd "$(dirname "$0")/../polee_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function $EXTRA -c vi.hip -o _obj/vi.o
touch _obj/*.o
make -s libpolee_hip.so libpolee_hip_untuned.so
