#!/bin/bash
# Same-card A/B of the headline sweep (task 1): the round-1 and round-2 trees (unpacked and built under _ab/r01, _ab/r02 by
#   git archive <rev> | tar -x -C _ab/rNN && (cd _ab/rNN && python -m rustsolver_amd.build)
# ), HEAD with the plain fill, HEAD with the saturating fill, one process per leg, interleaved ROUNDS times on ONE card.
# usage: tools/headline_ab.sh OUTDIR [ROUNDS]
set -e
out=${1:-gpurun_out/r04_ab}; rounds=${2:-5}
mkdir -p $out
export RS_JIT_CACHE=$PWD/$out/jitcache
: > $out/legs.jsonl
for r in $(seq 1 $rounds); do
  for leg in r01 r02 head_plain head_sat; do
    case $leg in
      r01) python tools/headline_ab.py --root _ab/r01 --label r01 ;;
      r02) python tools/headline_ab.py --root _ab/r02 --label r02 ;;
      head_plain) python tools/headline_ab.py --root . --label head_plain --saturating 0 ;;
      head_sat) python tools/headline_ab.py --root . --label head_sat --saturating 100 ;;
    esac >> $out/legs.jsonl
    echo "round $r $leg done"
  done
done
python - $out/legs.jsonl <<'PY'
import json, statistics, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
labels = []
for r in rows:
    if r["label"] not in labels:
        labels.append(r["label"])
print("| leg | avg launch ms (median of rounds; min..max) | algorithmic GB/s | plain copy GB/s on this card | kernel / copy | ms per step (wall) |")
print("|---|---|---|---|---|---|")
for lb in labels:
    rs = [r for r in rows if r["label"] == lb]
    m = lambda k: statistics.median(r[k] for r in rs)
    print("| %s | %.4f (%.4f..%.4f) | %.0f | %.0f | %.3f | %.4f |" % (lb, m("avg_launch_ms"), min(r["avg_launch_ms"] for r in rs), max(r["avg_launch_ms"] for r in rs),
                                                        m("algo_GBps"), m("copy_GBps"), m("over_copy"), m("ms_per_step")))
PY
