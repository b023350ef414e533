#!/usr/bin/env python
"""Condense the raw rocprofv3 output of scripts/collect_profiles.sh into profiles/<round>_<tag>_*.

    python scripts/profile_summary.py <tag> [round]     (reads gpurun_out/prof_<tag>/, writes profiles/)

Writes
  profiles/<round>_<tag>_kernel_stats.csv   the rocprofv3 --stats table (per-kernel calls / total / average ns)
  profiles/<round>_<tag>_summary.json       per kernel: launches, avg duration, HBM bytes per launch from the PMC
                                            passes (FETCH_SIZE doubled: gfx950 tallies the 128-B requests of wide
                                            coalesced reads at 64 B, MI355X_MICROARCH.md §HBM; WRITE_SIZE as is; both in
                                            KiB), MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel
                                            cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; plus the bench JSON line.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    if "nearest_kernel<true>" in name:                        # the evaluation-counting variant: one call per bench run, not the timed path
        return "nearest_kernel_counted"
    for k in ("igemm256h_kernel", "igemm128nh_kernel", "igemm128c_kernel"):      # patch-staged kernels: bf16 only, templated on the tap count
        if k in name:
            return k + "<bf16>"
    for k in ("igemm256_kernel", "igemm128n_kernel", "igemm_kernel", "wgrad256_kernel", "wgrad_slab_reduce_kernel",
              "conv_first_wgrad_kernel", "wgrad_kernel", "bn_apply_kernel", "bn_act_bwd_split_kernel", "bn_act_bwd_kernel", "bn_act_kernel", "maxpool2_act_bwd_kernel", "act_bwd_kernel", "case_prepare_kernel",
              "conv_first_kernel", "adam_kernel", "splitk_epilogue_kernel", "tapgemm",
              "tap_dw_kernel", "tap_dx_kernel", "transpose_weight_kernel", "maxpool2", "nearest_kernel", "points_pass_kernel",
              "case_setup_kernel", "bin_count_kernel", "bin_scan_kernel", "bin_scatter_kernel", "resample_init_kernel",
              "bn_stats_kernel", "sdf_"):
        if k in name:
            dt = "<bf16>" if ("DF16b" in name or "bf16" in name or "_Accum" in name) else ("<f32>" if ("If" in name or "float" in name) else "")
            return k + dt
    return name[:48]


def newest(pattern):
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


out = {"tag": tag, "kernels": {}}
# which tree the passes ran on: the digest collect_profiles.sh took ON THE BOX, and the commit (passed in through
# CFD_GIT_HEAD, else HEAD here at summary time, flagged when the work tree has uncommitted changes)
try:
    out.update(json.load(open(os.path.join(src, "source_stamp.json"))))
except Exception:
    pass
if not out.get("git_head"):
    import subprocess
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "cfd_amd", "bench.py"], cwd=root, capture_output=True, text=True).stdout.strip())
        out["git_head"] = head + ("+uncommitted" if dirty else "")
    except Exception:
        out["git_head"] = None
st = newest("stats/**/*_kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(st)):
        k = out["kernels"].setdefault(short(r["Name"]), {"launches": 0, "total_us": 0.0})
        k["launches"] += int(r["Calls"])
        k["total_us"] += float(r["TotalDurationNs"]) / 1e3
    for k in out["kernels"].values():
        k["avg_us"] = round(k["total_us"] / k["launches"], 2)
        k["total_us"] = round(k["total_us"], 1)


def pmc(sub):
    f = newest(f"{sub}/**/*_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write, mf = pmc("fetch"), pmc("write"), pmc("mfma")
for name, k in out["kernels"].items():
    if name in fetch and "FETCH_SIZE" in fetch[name]:
        v = fetch[name]["FETCH_SIZE"]
        k["hbm_read_bytes_per_launch"] = round(2.0 * 1024.0 * sum(v) / len(v))      # x2: gfx950 correction
    if name in write and "WRITE_SIZE" in write[name]:
        v = write[name]["WRITE_SIZE"]
        k["hbm_write_bytes_per_launch"] = round(1024.0 * sum(v) / len(v))
    if name in mf and "GRBM_GUI_ACTIVE" in mf[name]:
        busy, gui = sum(mf[name]["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(mf[name]["GRBM_GUI_ACTIVE"])
        if gui > 0:
            k["mfma_busy_frac"] = round(busy / (1024.0 * gui / 8.0), 4)
for which in ("stats", "fetch", "write", "mfma"):
    f = os.path.join(src, f"bench_{which}.json")
    if os.path.exists(f):
        try:
            out[f"bench_under_{which}"] = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            pass
out["kernels"] = dict(sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_us"]))
json.dump(out, open(os.path.join(dst, f"{rnd}_{tag}_summary.json"), "w"), indent=1)
for n, k in list(out["kernels"].items())[:12]:
    print(n, k)
