#!/usr/bin/env python3
"""Turns the output of tools/profile_deals_forms.sh (gpurun_out/<tag>/form{0,1}/...) into the committed before / after table of the deal path's kernels:

    python profiles/summarize_deals_forms.py gpurun_out/r03_deals_forms profiles/r03_deals

writes <prefix>.md and <prefix>_forms.json.  form0 = LDS delta tiles and one compaction job per root (RS_JIT_ROWS=0 RS_JIT_NO_SIBLINGS=1, the round-2 path), form1 = the engine's choice (delta rows in the list walkers).
Kernel times are per BATCH (both traversers' sweeps) from the serialised trace; `walks` = (deal, round subtree) pairs the batch's kernels of that class walked, as
rs_solver_walk_counts reports them, so that every class has a cost per walk.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
BATCHES = 7   # tools/time_three_street.py under the trace: 2 warm-up + 5 timed batches


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None


def classify(name):
    """kernel -> (class, round) of the three-street sweep"""
    n = re.sub(r"__[sg]\d+$", "", name.split("(")[0].replace("void ", ""))   # kernels of one form compiled together: one entry point per subtree shape
    if n.startswith("rs_tree_"):
        down = "_down" in n
        sparse = "_sparse" in n
        if not sparse:
            return ("flop reach-down" if down else "flop walk", 0)
        if down:
            return ("turn reach-down", 1)
        return ("turn walk" if n.endswith(("_pr", "_pr_wl")) else "river walk", 1 if n.endswith(("_pr", "_pr_wl")) else 2)
    for key, cls in (("k_compact", "live-deal lists (k_compact_live / k_compact_siblings)"), ("k_row_sums", "row sums (k_row_sums)"), ("k_worklist", "work lists (k_worklist)"),
                     ("k_build_shadow", "table shadow (k_build_shadow)"), ("k_apply_delta", "apply (k_apply_delta_jobs)"), ("k_deal_clusters", "dealing: get_cluster (k_deal_clusters)"),
                     ("k_deal_sample", "dealing: generate_hand + showdown (k_deal_sample)"), ("k_deal_prune_flags", "dealing: prune flags"), ("k_pack_attr", "per-deal records (k_pack_attr)"),
                     ("fillBuffer", "memset (counters, work lists)"), ("copyBuffer", "copies")):
        if key in n:
            return (cls, -1)
    return (n, -1)


def stats(form):
    f = one("%s/trace/**/*kernel_stats.csv" % form)
    out = collections.OrderedDict()
    names = collections.defaultdict(set)
    if not f:
        return out, names
    for r in csv.DictReader(open(f)):
        cls, _ = classify(r["Name"])
        e = out.setdefault(cls, {"ms_per_batch": 0.0, "launches_per_batch": 0.0})
        e["ms_per_batch"] += float(r["TotalDurationNs"]) / 1e6 / BATCHES
        e["launches_per_batch"] += float(r["Calls"]) / BATCHES
        names[cls].add(re.sub(r"__[sg]\d+$", "", r["Name"].split("(")[0].replace("void ", "")))
    return out, names


def pmc(form, sub):
    f = one("%s/%s/**/*counter_collection.csv" % (form, sub))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if not f:
        return acc, disp
    for r in csv.DictReader(open(f)):
        cls, _ = classify(r["Kernel_Name"])
        acc[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[cls].add(r["Dispatch_Id"])
    return acc, disp


def timing(form):
    out = {}
    for f in sorted(glob.glob(os.path.join(src, form, "time_*.log"))):
        txt = open(f).read()
        m = re.search(r"(\d+) deals per batch: ([0-9.]+) ms per batch = ([0-9.e+]+) deal-iterations/s", txt)
        w = re.search(r"traverser 0 \[(\d+), (\d+), (\d+)\] traverser 1 \[(\d+), (\d+), (\d+)\]", txt)
        if m:
            out[int(m.group(1))] = {"ms_per_batch": float(m.group(2)), "deal_iterations_per_s": float(m.group(3)),
                                    "walks_per_batch": [int(w.group(i)) + int(w.group(i + 3)) for i in (1, 2, 3)] if w else None}
    return out


forms = [open(os.path.join(src, d, "form.txt")).read().strip() for d in ("form0", "form1")]
st = [stats("form0"), stats("form1")]
tm = [timing("form0"), timing("form1")]
walks = (tm[1].get(4194304) or tm[0].get(4194304) or {}).get("walks_per_batch")
per_class_walks = {}
if walks:
    per_class_walks = {"flop walk": walks[0], "flop reach-down": walks[0], "turn walk": walks[1], "turn reach-down": walks[1], "river walk": walks[2],
                       "live-deal lists (k_compact_live / k_compact_siblings)": walks[1] + walks[2], "row sums (k_row_sums)": walks[1] + walks[2]}
L = ["# deal path, three streets, 5 000-bucket files: LDS delta tiles (round 2) against delta rows + sibling compaction (round 3)", "",
     "`tools/profile_deals_forms.sh`; form 0 = `%s`, form 1 = `%s` (the engine's own choice: delta rows and sibling compaction beyond 512 K deals per batch, strategy records and the reach-down hand-off at every size).  One MI355X." % (forms[0], forms[1]), "",
     "## batch time (hipGraph replay, the launches of a round overlapped on four streams: what `bench.py` measures)", "",
     "| deals per batch | round-2 forms: ms per batch | deal-iterations/s | round-3 forms: ms per batch | deal-iterations/s | ratio |", "|---|---|---|---|---|---|"]
for n in sorted(set(tm[0]) | set(tm[1]), reverse=True):
    a, b = tm[0].get(n), tm[1].get(n)
    L.append("| %d | %s | %s | %s | %s | %s |" % (n, "%.2f" % a["ms_per_batch"] if a else "-", "%.3g" % a["deal_iterations_per_s"] if a else "-", "%.2f" % b["ms_per_batch"] if b else "-",
                                                "%.3g" % b["deal_iterations_per_s"] if b else "-", "%.2fx" % (a["ms_per_batch"] / b["ms_per_batch"]) if a and b else "-"))
L += ["", "At 256 K and 64 K deals per batch both columns run the same kernel forms apart from the strategy records and the reach-down hand-off (the engine keeps tiles and one "
          "compaction job per root up to 512 K deals): their differences are the run-to-run spread of these small batches (several percent either way over the round's runs)."]
if walks:
    L += ["", "A 4 M-deal batch walks %s (deal, round subtree) pairs on flop / turn / river (both traversers; `rs_solver_walk_counts`): %.1f per deal." % (
        " / ".join("%.2f M" % (w / 1e6) for w in walks), sum(walks) / 4194304.0)]
L += ["", "## kernels, 4 M deals per batch, launches serialised (`RS_JIT_NO_OVERLAP=1`, no graph): ms per batch (launches per batch)", "",
      "| kernel class | round-2 forms | round-3 forms | walks per batch | round-3 forms: ps per walk |", "|---|---|---|---|---|"]
order = ["river walk", "turn walk", "flop walk", "turn reach-down", "flop reach-down", "row sums (k_row_sums)", "live-deal lists (k_compact_live / k_compact_siblings)", "work lists (k_worklist)",
         "table shadow (k_build_shadow)", "apply (k_apply_delta_jobs)", "per-deal records (k_pack_attr)", "dealing: get_cluster (k_deal_clusters)",
         "dealing: generate_hand + showdown (k_deal_sample)", "dealing: prune flags", "memset (counters, work lists)", "copies"]
seen = set()
tot = [0.0, 0.0]
for cls in order + [c for c in list(st[0][0]) + list(st[1][0]) if c not in order]:
    if cls in seen:
        continue
    seen.add(cls)
    a, b = st[0][0].get(cls), st[1][0].get(cls)
    if not a and not b:
        continue
    for i, e in enumerate((a, b)):
        tot[i] += e["ms_per_batch"] if e else 0.0
    w = per_class_walks.get(cls)
    L.append("| %s | %s | %s | %s | %s |" % (cls, "%.3f (%.0f)" % (a["ms_per_batch"], a["launches_per_batch"]) if a else "-",
                                          "%.3f (%.0f)" % (b["ms_per_batch"], b["launches_per_batch"]) if b else "-", "%.2f M" % (w / 1e6) if w else "",
                                          "%.0f" % (b["ms_per_batch"] * 1e9 / w) if (w and b) else ""))
L.append("| **sum of kernel times** | **%.2f** | **%.2f** | | |" % (tot[0], tot[1]))
L += ["", "Kernel names per class: " + "; ".join("%s = `%s`" % (c, "`, `".join(sorted(v))) for c, v in sorted(st[1][1].items()) if c.endswith("walk") or "down" in c), ""]
sq, sq_d = pmc("form1", "pmc_sq")
tcp, tcp_d = pmc("form1", "pmc_tcp")
if sq or tcp:
    L += ["## counters of the rows form (one batch, per DISPATCH of the class)", "",
          "| kernel class | dispatches | waves | VALU / wave | VMEM reads / wave | issue | stall | wait | L1 accesses | L1 accesses per VMEM read | L1->L2 reads | L1->L2 writes | TA data-stall cycles per CU |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for cls in order:
        if cls not in sq and cls not in tcp:
            continue
        n = max(1, len(sq_d.get(cls, ())))
        nt = max(1, len(tcp_d.get(cls, ())))
        s, t = sq.get(cls, {}), tcp.get(cls, {})
        waves = s.get("SQ_WAVES", 0.0)
        wc = s.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        vm = s.get("SQ_INSTS_VMEM_RD", 0.0)
        L.append("| %s | %d | %.0f | %.0f | %.1f | %.2f | %.2f | %.2f | %.3g | %.1f | %.3g | %.3g | %.3g |" % (
            cls, n, waves / n, s.get("SQ_INSTS_VALU", 0.0) / max(waves, 1.0), vm / max(waves, 1.0), s.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, s.get("SQ_WAIT_INST_ANY", 0.0) / wc,
            s.get("SQ_WAIT_ANY", 0.0) / wc, t.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / nt, (t.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / nt) / max(vm / n, 1.0),
            t.get("TCP_TCC_READ_REQ_sum", 0.0) / nt, t.get("TCP_TCC_WRITE_REQ_sum", 0.0) / nt, t.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0.0) / nt / 256.0))
    L.append("")
open(prefix + ".md", "w").write("\n".join(L) + "\n")
json.dump({"forms": forms, "timing": tm, "kernels_ms_per_batch": [{k: v for k, v in s[0].items()} for s in st], "walks_per_batch_4m": walks}, open(prefix + "_forms.json", "w"), indent=1)   # not <prefix>.json: bench.py reads profiles/r*_deals.json as summarize_deals.py output
print("\n".join(L[:40]))
