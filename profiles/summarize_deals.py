#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_deals.sh (gpurun_out/<tag>/<workload>/{trace,pmc_sq_a,pmc_sq_b,pmc_fetch,pmc_write}) into the committed
deal-path summary: per kernel and workload the dispatch time, waves, VALU / SALU / LDS / VMEM instruction counts, the share of wave cycles spent issuing,
stalled at issue and parked on a wait (SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES: disjoint, MI355X_MICROARCH.md), LDS bank-conflict
cycles and HBM bytes (FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE; KiB).

    python profiles/summarize_deals.py gpurun_out/r02_deals profiles/r02_deals
writes <prefix>.md and <prefix>.json (read by bench.py for `roofline_deals`).
"""
import collections
import csv
import glob
import json
import os
import sys

src, prefix = sys.argv[1], sys.argv[2]
MHZ = 2400.0   # MI355X peak engine clock; SQ cycle counters are in quad-cycles per the guide, ratios between them do not depend on it


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return g[-1] if g else None


def mine(kernel):
    return kernel.startswith(("void rs::", "rs::", "rs_tree_"))


def short(kernel):
    k = kernel.replace("void ", "")
    return k.split("(")[0]


out_json, lines = {}, ["# deal-path profile (%s)" % os.path.basename(prefix), "",
                       "rocprofv3 --kernel-trace (times) and --pmc passes (counters) of tools/time_deal_trainer.py / tools/time_three_street.py; SQ counters summed over all "
                       "dispatches of a kernel in the pass, then divided by its dispatches. `issue` / `stall` / `wait` = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over "
                       "SQ_WAVE_CYCLES (disjoint shares of the waves' lifetime: issuing, blocked at issue, parked on s_waitcnt or a barrier).", ""]
for wl in sorted(os.listdir(src)):
    d = os.path.join(src, wl)
    if not os.path.isdir(d):
        continue
    trace = one(wl + "/trace/*/*_kernel_trace.csv")
    times = collections.defaultdict(list)
    if trace:
        for r in csv.DictReader(open(trace)):
            if mine(r["Kernel_Name"]):
                times[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    counters = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(lambda: collections.defaultdict(int))
    meta = {}
    for p in ("pmc_sq_a", "pmc_sq_b", "pmc_fetch", "pmc_write"):
        f = one(wl + "/" + p + "/*/*_counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if not mine(r["Kernel_Name"]):
                continue
            k = short(r["Kernel_Name"])
            counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[k][r["Counter_Name"]] += 1
            meta[k] = (r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Workgroup_Size", "?"), r.get("Scratch_Size", "?"))
    log = one(wl + "/trace.log")
    headline = open(log).read().strip().splitlines()[-1] if log else ""
    lines += ["## %s" % wl, "", "`%s`" % headline, "",
              "| kernel | dispatches (trace) | avg us | total ms | VGPRs / LDS B / WG | waves / dispatch | VALU insts / wave | SALU / wave | LDS insts / wave | VMEM rd+wr / wave | issue | stall | wait | LDS conflict / LDS active | HBM bytes / dispatch |",
              "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    out_json[wl] = {"headline": headline, "kernels": {}}
    for k, v in sorted(times.items(), key=lambda kv: -sum(kv[1])):
        c, n = counters.get(k, {}), ndisp.get(k, {})

        def per(name):
            return c[name] / n[name] if n.get(name) else None
        waves = per("SQ_WAVES")
        wc = c.get("SQ_WAVE_CYCLES", 0.0)

        def perwave(name):
            return (c[name] / c["SQ_WAVES"]) if c.get("SQ_WAVES") and name in c else None

        def share(name):
            return c[name] / wc if wc and name in c else None
        hbm = None
        if n.get("FETCH_SIZE") or n.get("WRITE_SIZE"):
            hbm = (per("FETCH_SIZE") or 0.0) * 2 * 1024 + (per("WRITE_SIZE") or 0.0) * 1024
        conf = (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None
        vm = None
        if perwave("SQ_INSTS_VMEM_RD") is not None:
            vm = perwave("SQ_INSTS_VMEM_RD") + (perwave("SQ_INSTS_VMEM_WR") or 0.0)
        f = lambda x, fmt="%.3g": "-" if x is None else fmt % x
        m = meta.get(k, ("?", "?", "?", "?"))
        lines.append("| `%s` | %d | %.1f | %.2f | %s / %s / %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, m[0], m[1], m[2], f(waves, "%.0f"), f(perwave("SQ_INSTS_VALU"), "%.0f"), f(perwave("SQ_INSTS_SALU"), "%.0f"),
            f(perwave("SQ_INSTS_LDS"), "%.0f"), f(vm, "%.0f"), f(share("SQ_ACTIVE_INST_ANY"), "%.2f"), f(share("SQ_WAIT_INST_ANY"), "%.2f"), f(share("SQ_WAIT_ANY"), "%.2f"),
            f(conf, "%.3f"), f(hbm, "%.4g")))
        out_json[wl]["kernels"][k] = {"dispatches": len(v), "avg_us": sum(v) / len(v) / 1e3, "total_ms": sum(v) / 1e6, "waves_per_dispatch": waves,
                                      "valu_insts_per_wave": perwave("SQ_INSTS_VALU"), "salu_insts_per_wave": perwave("SQ_INSTS_SALU"), "lds_insts_per_wave": perwave("SQ_INSTS_LDS"),
                                      "vmem_insts_per_wave": vm, "issue_share": share("SQ_ACTIVE_INST_ANY"), "issue_stall_share": share("SQ_WAIT_INST_ANY"),
                                      "wait_share": share("SQ_WAIT_ANY"), "lds_conflict_over_active": conf, "hbm_bytes_per_dispatch": hbm,
                                      "vgprs": m[0], "lds_bytes": m[1], "workgroup": m[2], "scratch": m[3]}
    lines.append("")
open(prefix + ".md", "w").write("\n".join(lines) + "\n")
json.dump(out_json, open(prefix + ".json", "w"), indent=1)
print("\n".join(lines))
