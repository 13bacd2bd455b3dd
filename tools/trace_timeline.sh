#!/bin/bash
# Kernel timeline of the three-street deal trainer with its launches overlapped as in production (streams, no graph): start / end of every kernel of the last batch.
#   tools/trace_timeline.sh OUTDIR      (N = deals per batch, default 4 M)
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/$1"; mkdir -p "$OUT"
export RS_JIT_CACHE="$OUT/jitcache"
cd /tmp && export TMPDIR=/tmp
N=${N:-4194304} GRAPH=${GRAPH:-0} BATCHES=3 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$R/tools/time_three_street.py" > "$OUT/trace.log" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 - "$T" <<'PY' | tee "$OUT/timeline.txt"
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last batch: from the last-but-one k_build_shadow pair... take the last 2 sweeps = from the second-to-last "k_build_shadow" on
idx = [i for i, r in enumerate(rows) if "k_build_shadow" in r["Kernel_Name"]]
lo = idx[-2]
t0 = int(rows[lo]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n).replace("rs::", "").replace("void ", "")
    return re.sub(r"rs_tree_p[01]_", "", n)[:52]
busy_end = t0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %8.1f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
PY
find "$OUT" -name "*.db" -delete
