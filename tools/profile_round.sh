#!/bin/bash
# Runs on the GPU box (through gpurun): the three rocprofv3 passes that profiles/summarize_rocprof.py turns into profiles/<round>_*.
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh'
# Pass 1: kernel trace + stats of the default bench.py command.  Passes 2/3: FETCH_SIZE and WRITE_SIZE PMC counters, each in its
# own run (MI355X_MICROARCH.md, HBM section), on a shorter bench (5 steps, no CPU legs) -- the per-dispatch bytes do not depend on
# the step count.  rocprofv3 gets `python3` itself after `--` (no env / bash -c hop).
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/prof"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do echo "profile_round: still running"; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 50 --warmup 10 --skip solve_three_street > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace pass done" && tail -c 400 "$OUT/trace_bench.json" | head -c 200 && echo
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu --no-extra > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err"
echo "write pass done"
# keep the merge small: the per-dispatch traces of the PMC passes are enough, drop the sqlite / agent dumps
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
