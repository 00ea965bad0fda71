#!/bin/bash
# kernel-trace + stats only.  usage: tools/trace_bench.sh <tag> [bench args...]
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1
python3 "$REPO/tools/summarize_prof.py" "$OUT" | tee "$OUT/summary.txt"
