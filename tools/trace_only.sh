#!/bin/bash
# kernel trace only: tools/trace_only.sh <tag> <launches per step> [bench args]
TAG=$1; PER=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | cut -c1-300
python3 "$REPO/tools/trace_layers.py" "$OUT/trace" "$PER"
