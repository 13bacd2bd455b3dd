#!/bin/bash
# Runs on the GPU box (through gpurun): the three-street deal trainer (5 000-bucket files, 4 M deals per batch) under two kernel forms -- LDS delta tiles and one compaction job per root (RS_JIT_ROWS=0
# RS_JIT_NO_SIBLINGS=1: the round-2 path) and the engine's choice (delta rows in the list walkers) -- so that profiles/<round>_deals.md can put the kernels side by side.
#   gpurun --timeout 1100 -- 'bash tools/profile_deals_forms.sh TAG'
# Per form: the batch time as bench.py measures it (hipGraph replay, launches of one round overlapped), then a kernel trace with the launches serialised (RS_JIT_NO_OVERLAP=1:
# overlapped kernels stretch each other's durations) and, for the engine's choice, two PMC passes of one batch (--pmc with --kernel-trace only).
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-deals_forms}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do echo "profile_deals_forms: still running"; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
i=0
for form in "RS_JIT_ROWS=0 RS_JIT_NO_SIBLINGS=1" "default"; do
    if [ "$form" != default ]; then export $form; fi
    D="$OUT/form$i"; mkdir -p "$D"; echo "$form" > "$D/form.txt"
    for N in 4194304 1048576 262144 65536; do
        N=$N GRAPH=1 BATCHES=9 python3 "$R/tools/time_three_street.py" > "$D/time_$N.log" 2>&1
        echo "$form: $(grep three-street "$D/time_$N.log")"
    done
    RS_JIT_NO_OVERLAP=1 N=4194304 GRAPH=0 BATCHES=5 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 "$R/tools/time_three_street.py" > "$D/trace.log" 2> "$D/trace.err" || { tail -5 "$D/trace.err"; exit 1; }
    echo "$form serialised: $(grep three-street "$D/trace.log")"
    unset RS_JIT_ROWS RS_JIT_NO_SIBLINGS
    i=$((i+1))
done
D="$OUT/form1"
export RS_JIT_NO_OVERLAP=1 N=4194304 GRAPH=0 BATCHES=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$D/pmc_sq" -- python3 "$R/tools/time_three_street.py" > "$D/pmc_sq.log" 2> "$D/pmc_sq.err" || { tail -5 "$D/pmc_sq.err"; exit 1; }
echo "sq pass done"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d "$D/pmc_tcp" -- python3 "$R/tools/time_three_street.py" > "$D/pmc_tcp.log" 2> "$D/pmc_tcp.err" || { tail -5 "$D/pmc_tcp.err"; exit 1; }
echo "tcp pass done"
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*agent_info*" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
du -sh "$OUT"
