#!/usr/bin/env python3
"""Turns rocprofv3 output (gpurun_out/prof/{trace,pmc_fetch,pmc_write}) into the committed summaries.

    python profiles/summarize_rocprof.py gpurun_out/prof profiles/r01

Writes <prefix>_kernel_stats.csv (rocprofv3 --stats as-is), <prefix>_summary.md and
<prefix>_roofline_traffic.json (read back by bench.py for `roofline.traffic`).

Only dispatches with >= 512K work-items are summarised as "full-size": bench.py also times a
single-board trainer whose ~5 us launches would otherwise dominate the --stats averages.
PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate passes,
are in KiB, and on gfx950 FETCH_SIZE reports exactly half of a wide coalesced streaming read, so it is
doubled before being compared with byte counts.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
FULL = 1 << 19   # the LDS deal-batch kernels run 1024 workgroups of 512 threads


def kernel_source_sha16():
    """what the profiled kernels were generated from: bench.py recomputes it and reports the canned traffic as stale when the sources have moved on"""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("rs_device.hpp", "rs_jit.cpp", "rs_kernels.hip"):
        h.update(open(os.path.join(root, "rustsolver_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def git_head():
    import subprocess
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)), text=True).strip()
    except Exception:
        return None


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # gpurun MERGES runs into gpurun_out/: newest wins
    return g[-1] if g else None


stats = one("trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, prefix + "_kernel_stats.csv")

lines = ["# rocprofv3 summary (%s)" % os.path.basename(prefix), ""]
bench = one("trace_bench.json")
if bench:
    try:
        b = json.loads(open(bench).read().strip().splitlines()[-1])
        lines += ["bench.py under rocprofv3 --kernel-trace --stats: value = %.4g %s, ms_per_step = %.3f, "
                  "roofline.achieved = %.1f GB/s (HIP events, avg update launch %.1f us)"
                  % (b["value"], b["unit"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["avg_launch_ms"] * 1e3), ""]
    except Exception as e:  # noqa
        lines += ["(bench json unreadable: %s)" % e, ""]

trace = one("trace/*/*_kernel_trace.csv")
per_kernel = collections.defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace)):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        # the persistent deal kernels (resident LDS tiles) run ONE 512-thread workgroup per CU: 131 072 work-items for a 4 M-deal batch
        if (g >= FULL or ("_deals" in r["Kernel_Name"] and g >= (1 << 16))) and r["Kernel_Name"].startswith(("void rs::", "rs::", "rs_tree_")):
            per_kernel[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines += ["## full-size dispatches (>= 512K work-items), kernel trace", "",
              "| kernel | dispatches | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
    for k, v in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| `%s` | %d | %.1f | %.1f | %.1f | %.2f |" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, sum(v) / 1e6))
    upd = [d for k, v in per_kernel.items() if "k_update" in k for d in v]
    if upd:
        lines += ["", "all `k_update` full-size dispatches: %d, average %.1f us" % (len(upd), sum(upd) / len(upd) / 1e3), ""]
    tre = [d for k, v in per_kernel.items() if k in ("rs_tree_p0_lanes", "rs_tree_p1_lanes") for d in v]
    xr = [d for k, v in per_kernel.items() if k in ("rs_tree_p0_lanes_xr", "rs_tree_p1_lanes_xr") for d in v]
    if xr:
        lines += ["all `rs_tree_p*_lanes_xr` (config 3: river subtrees below ENUM chance nodes, 11.76 M lanes each) full-size dispatches: %d, average %.1f us"
                  % (len(xr), sum(xr) / len(xr) / 1e3), ""]
    if tre:
        lines += ["all `rs_tree_p*_lanes` (river tree, lane model) full-size dispatches: %d, average %.1f us"
                  % (len(tre), sum(tre) / len(tre) / 1e3), ""]

traffic = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = one(name + "/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if (int(r["Grid_Size"]) >= FULL or ("_deals" in r["Kernel_Name"] and int(r["Grid_Size"]) >= (1 << 16))) and r["Counter_Name"] == counter and ("rs::" in r["Kernel_Name"] or "rs_tree_" in r["Kernel_Name"]):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    traffic[counter] = agg
if traffic:
    lines += ["## HBM traffic from PMC (separate passes; KiB; FETCH_SIZE doubled per the gfx950 correction)", "",
              "| kernel | dispatches | FETCH_SIZE avg KiB (raw) | read bytes (x2 x1024) | WRITE_SIZE avg KiB | write bytes | total bytes / dispatch |",
              "|---|---|---|---|---|---|---|"]
    fam = {"update": [0.0, 0], "tree": [0.0, 0]}
    for k in sorted(set(traffic.get("FETCH_SIZE", {})) | set(traffic.get("WRITE_SIZE", {}))):
        fv, wv = traffic.get("FETCH_SIZE", {}).get(k, []), traffic.get("WRITE_SIZE", {}).get(k, [])
        fa = sum(fv) / len(fv) if fv else 0.0
        wa = sum(wv) / len(wv) if wv else 0.0
        total = fa * 2 * 1024 + wa * 1024
        lines.append("| `%s` | %d | %.0f | %.4g | %.0f | %.4g | %.4g |" % (k, max(len(fv), len(wv)), fa, fa * 2048, wa, wa * 1024, total))
        for name in fam:
            if ("k_" + name in k) or (name == "tree" and k in ("rs_tree_p0_lanes", "rs_tree_p1_lanes")):
                fam[name][0] += total * max(len(fv), len(wv))
                fam[name][1] += max(len(fv), len(wv))
    for name, (tot, n) in fam.items():
        if n:
            per_launch = tot / n
            lines += ["", "`k_%s`: PMC HBM bytes per dispatch (dispatch-weighted) = %.4g" % (name, per_launch)]
            json.dump({"kernel": "rs::k_" + name, "hbm_bytes_per_launch": per_launch, "dispatches": n, "kernel_source_sha16": kernel_source_sha16(), "commit": git_head(),
                       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), KiB x1024",
                       "workload": "bench.py default (9216 boards x 1000 clusters, clamp)"},
                      open(prefix + "_roofline_traffic_%s.json" % name, "w"), indent=1)
open(prefix + "_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
