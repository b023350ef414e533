#!/bin/bash
# usage: scripts/diag_clock.sh <mask>...  effective clock and MFMA-busy of U d5 (B = 128, forward) under each CFD_DIAG
# variant (scripts/diag_variants.sh), from one rocprofv3 --pmc pass per variant
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = 0 ]; then export CFD_AMD_LIB=""; else export CFD_AMD_LIB=$R/variants/libdiag$v.so; fi
  out=$R/gpurun_out/diagclk_$v
  rm -rf $out; mkdir -p $out
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 $R/scripts/layer_bench.py --net U --batch 128 --modes fwd --only d5 --bias 1 > $out/log.txt 2>&1 || exit 1
  python3 - "$out" "$v" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list); ns = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm256h" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": ns.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
gui = sum(acc["GRBM_GUI_ACTIVE"]) / 8; t = sum(ns)
print("diag %3s  launches %3d  avg %7.1f us  clock %.3f GHz  mfma busy %.3f" % (sys.argv[2], len(ns), t / len(ns) / 1e3, gui / t,
      sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * gui)))
PY
done
