#!/bin/bash
# Runs on the GPU box (through gpurun): puts the DEAL path (the algorithm the reference runs: sampled mccfr over deal batches, DESIGN.md sections 2a / 8a) on a measured footing.
#   gpurun --timeout 1100 -- 'bash tools/profile_deals.sh [TAG]'
# Per workload: one kernel-trace pass (times) and PMC passes with --kernel-trace only (no other trace domain), SQ counters at most 8 per pass, FETCH_SIZE and
# WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, "rocprofv3 PMC slots").  rocprofv3 gets `python3` itself after `--` (no env / bash -c hop): the
# workload's knobs are exported into this shell's environment instead.
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-deals}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do echo "profile_deals: still running"; done ) &
HEART=$!
trap "kill $HEART 2>/dev/null" EXIT
SQ_A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
SQ_B="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
run_workload() {   # name script
    local name="$1" script="$2"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/trace" -- python3 "$R/tools/$script" > "$OUT/$name/trace.log" 2> "$OUT/$name/trace.err" || { tail -5 "$OUT/$name/trace.err"; return 1; }
    echo "$name trace: $(tail -1 "$OUT/$name/trace.log")"
    BATCHES=3 rocprofv3 --kernel-trace --pmc $SQ_A --output-format csv -d "$OUT/$name/pmc_sq_a" -- python3 "$R/tools/$script" > "$OUT/$name/pmc_sq_a.log" 2> "$OUT/$name/pmc_sq_a.err" || { tail -5 "$OUT/$name/pmc_sq_a.err"; return 1; }
    BATCHES=3 rocprofv3 --kernel-trace --pmc $SQ_B --output-format csv -d "$OUT/$name/pmc_sq_b" -- python3 "$R/tools/$script" > "$OUT/$name/pmc_sq_b.log" 2> "$OUT/$name/pmc_sq_b.err" || { tail -5 "$OUT/$name/pmc_sq_b.err"; return 1; }
    BATCHES=3 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- python3 "$R/tools/$script" > "$OUT/$name/pmc_fetch.log" 2> "$OUT/$name/pmc_fetch.err" || { tail -5 "$OUT/$name/pmc_fetch.err"; return 1; }
    BATCHES=3 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- python3 "$R/tools/$script" > "$OUT/$name/pmc_write.log" 2> "$OUT/$name/pmc_write.err" || { tail -5 "$OUT/$name/pmc_write.err"; return 1; }
    echo "$name pmc passes done"
}
mkdir -p "$OUT/river_64k" "$OUT/river_4m" "$OUT/three_street_4m"
export GRAPH=0
N=65536 run_workload river_64k time_deal_trainer.py
N=4194304 run_workload river_4m time_deal_trainer.py
N=4194304 run_workload three_street_4m time_three_street.py
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*agent_info*" -delete
du -sh "$OUT"
