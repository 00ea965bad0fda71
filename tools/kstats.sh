#!/bin/bash
# GPU box: rocprofv3 kernel stats of `python3 <args>`; prints the top kernels.  usage: tools/kstats.sh <tag> <script> [args]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/r4/ks_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$REPO/$1" "${@:2}" > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {100*float(r['TotalDurationNs'])/tot:5.1f}")
PY
