#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_config3.sh into profiles/<prefix>.md: per kernel of the config-3 sweep (706-node tree, 5 000 clusters, boards 1/49/2 352)
the dispatches, average / total time per iteration and the HBM bytes per dispatch from the PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, KiB).

    python profiles/summarize_config3.py gpurun_out/r02_config3 profiles/r02_config3
"""
import collections
import csv
import glob
import os
import sys

src, prefix = sys.argv[1], sys.argv[2]


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return g[-1] if g else None


def short(k):
    return k.replace("void ", "").split("(")[0]


ITER = 11   # tools/time_config3.py: 2 warm-up + 3 x 3 timed iterations
times = collections.defaultdict(list)
for r in csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))):
    if r["Kernel_Name"].startswith(("void rs::", "rs::", "rs_tree_")):
        times[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = one(name + "/*/*_counter_collection.csv")
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    pmc[counter] = agg
lines = ["# config 3 at size under rocprofv3 (%s)" % os.path.basename(prefix), "", "`%s`" % open(one("trace.log")).read().strip().splitlines()[-1], "",
         "| kernel | dispatches / iteration | avg us | ms / iteration | PMC read bytes / dispatch (FETCH_SIZE x2 x1024) | PMC write bytes / dispatch | PMC GB / iteration | PMC TB/s |",
         "|---|---|---|---|---|---|---|---|"]
tot_ms = tot_gb = 0.0
for k, v in sorted(times.items(), key=lambda kv: -sum(kv[1])):
    if "fill" in k or "checksum" in k:
        continue
    f, w = pmc["FETCH_SIZE"].get(k, []), pmc["WRITE_SIZE"].get(k, [])
    rd = sum(f) / len(f) * 2048 if f else 0.0
    wr = sum(w) / len(w) * 1024 if w else 0.0
    ms_it = sum(v) / 1e6 / ITER
    gb_it = (rd + wr) * len(v) / ITER / 1e9
    tot_ms += ms_it
    tot_gb += gb_it
    lines.append("| `%s` | %.1f | %.1f | %.3f | %.4g | %.4g | %.2f | %.2f |" % (k, len(v) / ITER, sum(v) / len(v) / 1e3, ms_it, rd, wr, gb_it, (gb_it / ms_it) if ms_it else 0.0))
lines += ["", "all kernels: %.2f ms and %.1f GB of HBM traffic per iteration = %.2f TB/s" % (tot_ms, tot_gb, tot_gb / tot_ms)]
open(prefix + ".md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
