#!/usr/bin/env python3
"""Turns the output of tools/profile_deals_ab.sh (gpurun_out/<tag>/form<i>/{trace,sq_a,sq_b,tcp,tcp2,tcc,fetch,write}) into one markdown table per kernel form:
per kernel the dispatch time (kernel-trace pass), and per dispatch the counters of the PMC passes (sums over XCDs / SEs / CUs as rocprofv3 reports them).

    python profiles/summarize_deals_ab.py gpurun_out/r3ab profiles/r03_deals_ab.md
"""
import collections
import csv
import glob
import os
import sys

src, out_path = sys.argv[1], sys.argv[2]


def short(k):
    return k.replace("void ", "").split("(")[0]


lines = ["# deal path, three streets, 5 000-bucket files, 4 M deals per batch: kernel forms side by side (%s)" % os.path.basename(src), "",
         "`tools/profile_deals_ab.sh`: hipGraph off, 3 timed batches per PMC pass, one pass per counter group; every value is per DISPATCH of the kernel "
         "(sum over the pass / dispatches in the pass).  FETCH_SIZE is doubled (gfx950: the counter reports 32-byte units as 64), bytes = KiB x 1024.", ""]
for form in sorted(glob.glob(os.path.join(src, "form*"))):
    name = open(os.path.join(form, "form.txt")).read().strip()
    tl = open(os.path.join(form, "trace.log")).read().strip().splitlines()[-1]
    lines += ["## %s" % name, "", "`%s`" % tl, ""]
    times = {}
    for f in glob.glob(os.path.join(form, "trace", "*", "*kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            times[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6)
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    for p in ("sq_a", "sq_b", "tcp", "tcp2", "tcc", "fetch", "write"):
        for f in glob.glob(os.path.join(form, p, "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                ctr[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k][row["Counter_Name"]].add(row["Dispatch_Id"])
                meta[k] = (row["VGPR_Count"], row["LDS_Block_Size"], row["Workgroup_Size"])
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"]
    lines += ["| kernel | calls | avg us | total ms | VGPR / LDS / WG | waves | VALU / wave | SALU / wave | SMEM / wave | VMEM rd+wr / wave | LDS / wave | issue | stall | wait | "
              "L1->L2 rd req | rd latency (cyc) | L1 accesses | TCP pending-stall / busy wave-cycles | L1->L2 wr req | atomics (ret / no-ret) | L2 req | L2 hit | L2 atomics | fetch MB | write MB |",
              "|" + "---|" * 25]
    for k, (calls, avg, tot) in sorted(times.items(), key=lambda kv: -kv[1][2]):
        if not (k.startswith("rs::") or k.startswith("rs_tree_")) or tot < 0.3:
            continue
        c = ctr.get(k, {})

        def per(n):
            d = len(disp[k][n]) if k in disp and n in disp[k] else 0
            return c[n] / d if d else None
        w = per("SQ_WAVES")
        wc = per("SQ_WAVE_CYCLES")

        def pw(n):
            v = per(n)
            return "%.0f" % (v / w) if (v is not None and w) else "-"

        def sh(n):
            v = per(n)
            return "%.2f" % (v / wc) if (v is not None and wc) else "-"
        rd, lat = per("TCP_TCC_READ_REQ_sum"), per("TCP_TCC_READ_REQ_LATENCY_sum")
        vm = (per("SQ_INSTS_VMEM_RD") or 0) + (per("SQ_INSTS_VMEM_WR") or 0)
        hit, req = per("TCC_HIT_sum"), per("TCC_REQ_sum")
        miss = per("TCC_MISS_sum")
        f_, wr_ = per("FETCH_SIZE"), per("WRITE_SIZE")
        pend = per("TCP_PENDING_STALL_CYCLES_sum")

        def g(v, fmt="%.3g"):
            return fmt % v if v is not None else "-"
        lines.append("| `%s` | %d | %.1f | %.2f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s / %s | %s | %s | %s | %s | %s |" % (
            k, calls, avg, tot, " / ".join(meta.get(k, ("-", "-", "-"))), g(w, "%.0f"), pw("SQ_INSTS_VALU"), pw("SQ_INSTS_SALU"), pw("SQ_INSTS_SMEM"),
            ("%.0f" % (vm / w)) if w else "-", pw("SQ_INSTS_LDS"), sh("SQ_ACTIVE_INST_ANY"), sh("SQ_WAIT_INST_ANY"), sh("SQ_WAIT_ANY"),
            g(rd), ("%.0f" % (lat / rd)) if (rd and lat) else "-", g(per("TCP_TOTAL_CACHE_ACCESSES_sum")),
            ("%.2f" % (pend / (wc * 4))) if (pend is not None and wc) else "-", g(per("TCP_TCC_WRITE_REQ_sum")),
            g(per("TCP_TCC_ATOMIC_WITH_RET_REQ_sum")), g(per("TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum")), g(req),
            ("%.2f" % (hit / (hit + miss))) if (hit is not None and miss is not None and hit + miss > 0) else "-", g(per("TCC_ATOMIC_sum")),
            ("%.1f" % (f_ * 2 * 1024 / 1e6)) if f_ is not None else "-", ("%.1f" % (wr_ * 1024 / 1e6)) if wr_ is not None else "-"))
    lines.append("")
open(out_path, "w").write("\n".join(lines) + "\n")
print("wrote", out_path)
