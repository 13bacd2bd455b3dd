#!/bin/bash
# Kernel trace of the three-street deal trainer (4 M deals per batch, launches serialised) under a list of environments: per-kernel-class times side by side.
#   tools/profile_walks.sh OUTDIR "ENV1" "ENV2" ...      ("default" = nothing set)
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/$1"; shift
mkdir -p "$OUT"
export RS_JIT_CACHE="$OUT/jitcache"
cd /tmp && export TMPDIR=/tmp
i=0
for form in "$@"; do
    D="$OUT/form$i"; rm -rf "$D"; mkdir -p "$D"; echo "$form" > "$D/form.txt"
    (
    if [ "$form" != default ]; then export $form; fi
    N=${N:-4194304} GRAPH=1 BATCHES=9 python3 "$R/tools/time_three_street.py" > "$D/time.log" 2>&1 || { tail -5 "$D/time.log"; exit 1; }
    echo "$form: $(grep three-street "$D/time.log")"
    RS_JIT_NO_OVERLAP=1 N=${N:-4194304} GRAPH=0 BATCHES=5 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 "$R/tools/time_three_street.py" > "$D/trace.log" 2> "$D/trace.err" || { tail -5 "$D/trace.err"; exit 1; }
    )
    S=$(find "$D/trace" -name "*kernel_stats.csv" | head -1)
    python3 - "$S" <<'PY' | tee "$D/classes.txt"
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
cls = {}
for r in rows:
    n = r["Name"]
    if n.startswith("rs_tree"):
        k = re.sub(r"rs_tree_p[01]_", "", n)
        k = re.sub(r"__s\d+$", "", k)
    else:
        k = n.split("(")[0][:60]
    c = cls.setdefault(k, [0, 0.0])
    c[0] += int(r["Calls"]); c[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in cls.values())
for k, v in sorted(cls.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-70s calls %6d  ms/batch %8.3f" % (k, v[0], v[1] / 1e6 / 7))   # 2 warm-up + 5 timed batches
print("  total kernel ms/batch %.3f" % (tot / 1e6 / 7))
PY
    find "$D" -name "*.db" -delete; find "$D" -name "*agent_info*" -delete; find "$D" -name "*kernel_trace.csv" -delete
    i=$((i+1))
done
