#!/bin/bash
# SQ counter passes only.  usage: tools/pmc_sq.sh <tag> [bench args...]
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --prime-ms 10 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
python3 "$REPO/tools/summarize_prof.py" "$OUT" | tee "$OUT/summary.txt"
tail -3 "$OUT/pmc_sq2.log"
