#!/bin/bash
# usage: scripts/diag_variants.sh <mask>...   build variants/libdiag<mask>.so: the library with igemm256h.hip compiled
# under -DCFD_DIAG=<mask> (pieces of the K loop taken away: see the comment at CFD_DIAG in igemm256h.hip).  For
# `CFD_AMD_LIB=variants/libdiag<mask>.so python scripts/layer_bench.py ...` timing runs only: results are garbage.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
python -m cfd_amd.build > /dev/null
mkdir -p "$root/variants"
for m in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -fno-gpu-rdc -DCFD_DIAG=$m \
    -c "$root/cfd_amd/csrc/igemm256h.hip" -o "$root/variants/igemm256h_diag$m.o"
  objs=$(ls "$root"/cfd_amd/lib/*.o | grep -v igemm256h.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o "$root/variants/libdiag$m.so" $objs "$root/variants/igemm256h_diag$m.o"
  echo "$root/variants/libdiag$m.so"
done
