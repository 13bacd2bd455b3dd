#!/bin/bash
# Same-card A/B of the three-street deal trainer (tools/time_three_street.py: 5 000-bucket files, hipGraph replay) between a baseline tree unpacked and built under _ab/base
# (git archive <rev> | tar -x -C _ab/base && (cd _ab/base && python -m rustsolver_amd.build)) and this tree, one process per leg, interleaved.
# usage: tools/deals_ab.sh OUTDIR [ROUNDS] [SIZES]      environment for the HEAD legs only: HEAD_ENV="RS_JIT_X=1 ..."
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=$R/${1:-gpurun_out/deals_ab}; rounds=${2:-3}; sizes=${3:-"4194304 1048576 65536"}
mkdir -p $out
export RS_JIT_CACHE=$out/jitcache
: > $out/lines.txt
for r in $(seq 1 $rounds); do
  for N in $sizes; do
    for leg in base head; do
      if [ $leg = base ]; then dir=$R/_ab/base; extra=""; else dir=$R; extra="$HEAD_ENV"; fi
      line=$(cd $dir && env $extra N=$N GRAPH=1 BATCHES=9 python3 tools/time_three_street.py 2>$out/err_$leg.log | grep three-street) || { tail -5 $out/err_$leg.log; exit 1; }
      echo "$leg $N $line" | tee -a $out/lines.txt
    done
  done
done
