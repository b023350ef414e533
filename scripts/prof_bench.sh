#!/bin/bash
# usage: scripts/prof_bench.sh <tag> <bench.py args...>: rocprofv3 kernel stats of one bench.py run -> gpurun_out/prof_<tag>/
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
