#!/bin/bash
# usage: scripts/ab_bench.sh <tag> <rounds> <workload> <envA> <envB>   (env = "VAR=value [VAR=value ...]" or "")
# interleaved A/B bench runs on ONE box (box-to-box spread is +-3 %, larger than most single changes)
# -> gpurun_out/ab_<tag>.txt, one "A|B ms_per_step frac" line per run
tag=$1; rounds=$2; wl=$3; ea=$4; eb=$5
out=gpurun_out/ab_$tag.txt; : > $out
for r in $(seq $rounds); do
  for arm in A B; do
    e=$ea; [ $arm = B ] && e=$eb
    env $e timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --no-also > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || exit 1
    python -c "import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$arm', d['ms_per_step'], d['roofline']['frac'])" >> $out
  done
done
cat $out
