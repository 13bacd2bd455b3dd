#!/bin/bash
# Kernel trace + PMC passes (rocprofv3 --pmc with --kernel-trace only) over tools/time_best_response.py (full 1 176-combo ranges unless HANDS is set): per kernel, calls and time
# per exploitability call, waves, issue / wait shares, L1 accesses per vector-memory read, L2 hit rate, HBM bytes fetched and written (FETCH_SIZE x2 on gfx950, KiB).
#   tools/profile_br_pmc.sh OUTDIR
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/$1"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/time_best_response.py" > "$OUT/trace.log" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/tools/time_best_response.py" > "$OUT/pmc_sq.log" 2> "$OUT/pmc_sq.err" || { tail -5 "$OUT/pmc_sq.err"; exit 1; }
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcp" -- python3 "$R/tools/time_best_response.py" > "$OUT/pmc_tcp.log" 2> "$OUT/pmc_tcp.err" || { tail -5 "$OUT/pmc_tcp.err"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/tools/time_best_response.py" > "$OUT/pmc_fetch.log" 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/tools/time_best_response.py" > "$OUT/pmc_write.log" 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/summary.md"
import csv, glob, sys, collections
out = sys.argv[1]
cls = lambda n: n.split("(")[0].replace("void ", "")[:44]
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
tm = {}
for r in csv.DictReader(open(stats[0])):
    tm[cls(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("pmc_sq", "pmc_tcp", "pmc_fetch", "pmc_write"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[cls(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
print(open(out + "/trace.log").read().strip().splitlines()[-1])
print()
print("| kernel | launches per call | ms per call | waves per launch | VALU/wave | VMEM rd/wave | LDS/wave | issue | wait | L1 acc / VMEM rd | L2 hit rate | HBM GB read per call | HBM GB written per call |")
print("|" + "---|" * 13)
calls = 4.0   # tools/time_best_response.py makes four exploitability calls
for k, (n, ns) in sorted(tm.items(), key=lambda kv: -kv[1][1])[:8]:
    c = agg.get(k, {})
    g = lambda x: c.get(x, 0.0)
    w = g("SQ_WAVES") or 1.0
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    print("| `%s` | %.0f | %.1f | %.0f | %.0f | %.1f | %.1f | %.2f | %.2f | %.1f | %.2f | %.2f | %.2f |" % (
        k, n / calls, ns / 1e6 / calls, w / max(1, n), g("SQ_INSTS_VALU") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_LDS") / w,
        g("SQ_ACTIVE_INST_ANY") / max(1, g("SQ_WAVE_CYCLES")), g("SQ_WAIT_ANY") / max(1, g("SQ_WAVE_CYCLES")),
        g("TCP_TOTAL_CACHE_ACCESSES_sum") / max(1, g("SQ_INSTS_VMEM_RD")), hit / max(1, hit + miss),
        g("FETCH_SIZE") * 2 * 1024 / 1e9 / calls, g("WRITE_SIZE") * 1024 / 1e9 / calls))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*agent_info*" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +20M -delete
