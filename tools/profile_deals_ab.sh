#!/bin/bash
# Runs on the GPU box (through gpurun): the three-street deal sweep's kernels under the memory-pipeline counters of tools/profile_deals_deep.sh, once per kernel form, so that
# two forms can be read side by side (round 3: LDS-tile list walkers against the ordered / segment-summing ones).
#   gpurun --timeout 1100 -- 'bash tools/profile_deals_ab.sh TAG "RS_JIT_ORDERED=0" "RS_JIT_ORDERED=1"'
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="$1"; shift
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do echo "profile_deals_ab: still running"; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
export N=4194304 GRAPH=0 BATCHES=3
pass() {   # form-dir name counters...
    local dir="$1" name="$2"; shift; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$dir/$name" -- python3 "$R/tools/time_three_street.py" > "$dir/$name.log" 2> "$dir/$name.err" || { tail -5 "$dir/$name.err"; return 1; }
    echo "$name done: $(tail -1 "$dir/$name.log")"
}
i=0
for form in "$@"; do
    export $form
    D="$OUT/form$i"; mkdir -p "$D"; echo "$form" > "$D/form.txt"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 "$R/tools/time_three_street.py" > "$D/trace.log" 2> "$D/trace.err" || { tail -5 "$D/trace.err"; exit 1; }
    echo "form $i ($form) trace: $(tail -1 "$D/trace.log")"
    pass "$D" sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
    pass "$D" sq_b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_WAIT_INST_LDS
    pass "$D" tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
    pass "$D" tcp2 TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_GATE_EN1_sum
    pass "$D" tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
    pass "$D" fetch FETCH_SIZE
    pass "$D" write WRITE_SIZE
    for v in ${form}; do unset ${v%%=*}; done
    i=$((i+1))
done
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*agent_info*" -delete
find "$OUT" -name "*kernel_trace.csv" -size +30M -delete
du -sh "$OUT"
