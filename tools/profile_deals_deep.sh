#!/bin/bash
# Runs on the GPU box (through gpurun): a second look at the three-street deal sweep's UP kernels -- memory-pipeline counters this time (vector-memory and LDS latency as
# seen by the SQ, L1 / TLB behaviour, L2 hit rate and atomics, workgroup-launch stalls on LDS).  One rocprofv3 --pmc pass per counter group, --kernel-trace only.
#   gpurun --timeout 900 -- 'bash tools/profile_deals_deep.sh [TAG]'
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-deals_deep}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do echo "profile_deals_deep: still running"; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
export N=4194304 GRAPH=0 BATCHES=3
pass() {   # name counters...
    local name="$1"; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/tools/time_three_street.py" > "$OUT/$name.log" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; return 1; }
    echo "$name done: $(tail -1 "$OUT/$name.log")"
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC
pass tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_ACCESSES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
pass spi SPI_RA_LDS_CU_FULL_CSN SPI_CSN_BUSY SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_NUM_THREADGROUPS
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*agent_info*" -delete
du -sh "$OUT"
