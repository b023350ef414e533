#!/bin/bash
# usage: scripts/build_variant.sh <git-ref> <name>: build the library as it was at <git-ref> into variants/lib<name>.so
# (for interleaved A/B runs on one box: CFD_AMD_LIB=<that file> makes cfd_amd/_lib.py load it instead of the in-tree build).
# Only csrc/ and include/ come from <git-ref>; the Python side is the working tree's, so <git-ref> must export the same C ABI.
set -e
ref=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/cfd_amd" "$root/variants"
git -C "$root" archive "$ref" cfd_amd/csrc include cfd_amd/build.py | tar -x -C "$tmp"
touch "$tmp/cfd_amd/__init__.py"
(cd "$tmp" && python -c "
import sys; sys.path.insert(0, '.')
from cfd_amd import build
print(build.build(force=True))")
cp "$tmp/cfd_amd/lib/libcfd_amd.so" "$root/variants/lib$name.so"
rm -rf "$tmp"
echo "$root/variants/lib$name.so"
