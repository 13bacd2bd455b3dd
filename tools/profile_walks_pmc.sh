#!/bin/bash
# PMC passes (rocprofv3 --pmc with --kernel-trace only) over ONE 4 M-deal batch of the three-street trainer, launches serialised; per kernel class: waves, issue / wait shares,
# L1 accesses per vector-memory read, L1<->L2 requests, L2 hits and misses.
#   tools/profile_walks_pmc.sh OUTDIR ["ENV"]
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/$1"; form="${2:-default}"
mkdir -p "$OUT"
export RS_JIT_CACHE="$OUT/jitcache"
cd /tmp && export TMPDIR=/tmp
if [ "$form" != default ]; then export $form; fi
export RS_JIT_NO_OVERLAP=1 N=${N:-4194304} GRAPH=0 BATCHES=1
python3 "$R/tools/time_three_street.py" > "$OUT/warm.log" 2>&1   # fills the kernel cache
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/tools/time_three_street.py" > "$OUT/pmc_sq.log" 2> "$OUT/pmc_sq.err" || { tail -5 "$OUT/pmc_sq.err"; exit 1; }
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d "$OUT/pmc_tcp" -- python3 "$R/tools/time_three_street.py" > "$OUT/pmc_tcp.log" 2> "$OUT/pmc_tcp.err" || { tail -5 "$OUT/pmc_tcp.err"; exit 1; }
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/pmc_tcc" -- python3 "$R/tools/time_three_street.py" > "$OUT/pmc_tcc.log" 2> "$OUT/pmc_tcc.err" || { tail -5 "$OUT/pmc_tcc.err"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 "$R/tools/time_three_street.py" > "$OUT/pmc_sq2.log" 2> "$OUT/pmc_sq2.err" || echo "sq2 pass failed (counters?)"
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
def cls(n):
    if n.startswith("rs_tree"):
        return re.sub(r"__s\d+$", "", re.sub(r"rs_tree_p[01]_", "", n))
    return n.split("(")[0][:48]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in ("pmc_sq", "pmc_tcp", "pmc_tcc", "pmc_sq2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            k = cls(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
print("| kernel class | dispatches | waves | VALU/wave | VMEM rd/wave | VMEM wr/wave | LDS/wave | issue | wait | L1 acc / VMEM rd | L1->L2 rd req | L1->L2 wr req | L2 hit | L2 miss | L2 hit rate | EA rd req | EA wr req | TA stall cyc |")
print("|" + "---|" * 19)
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVES", 0) or 1
    g = lambda n: c.get(n, 0.0)
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    print("| %s | %d | %d | %.0f | %.1f | %.1f | %.1f | %.2f | %.2f | %.1f | %.3g | %.3g | %.3g | %.3g | %.2f | %.3g | %.3g | %.3g |" % (
        k, len(disp[k]) // 3 or len(disp[k]), w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_VMEM_WR") / w, g("SQ_INSTS_LDS") / w,
        g("SQ_ACTIVE_INST_ANY") / max(1, g("SQ_WAVE_CYCLES")), g("SQ_WAIT_ANY") / max(1, g("SQ_WAVE_CYCLES")),
        g("TCP_TOTAL_CACHE_ACCESSES_sum") / max(1, g("SQ_INSTS_VMEM_RD")), g("TCP_TCC_READ_REQ_sum"), g("TCP_TCC_WRITE_REQ_sum"), hit, miss, hit / max(1, hit + miss),
        g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_WRREQ_sum"), g("TCP_TCP_TA_DATA_STALL_CYCLES_sum")))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*agent_info*" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
