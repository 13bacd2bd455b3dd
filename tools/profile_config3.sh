#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes of BASELINE configs[2] at size (tools/time_config3.py: 706-node tree, 5 000 clusters, boards 1/49/2 352, 135 GB):
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md).  `python3` itself after `--`.
#   gpurun --timeout 1100 -- 'bash tools/profile_config3.sh [TAG]'
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-prof_config3}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do echo "profile_config3: still running"; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/time_config3.py" > "$OUT/trace.log" 2> "$OUT/trace.err"
echo "trace: $(tail -1 "$OUT/trace.log")"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/tools/time_config3.py" > "$OUT/pmc_fetch.log" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/tools/time_config3.py" > "$OUT/pmc_write.log" 2> "$OUT/pmc_write.err"
echo "write pass done"
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*agent_info*" -delete
du -sh "$OUT"
