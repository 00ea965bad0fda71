#!/bin/bash
# GPU box: same-box A/B of one script under different environments (in-tree library).
# usage: tools/ab_env.sh "<script args>" "VAR=a" "VAR=b" ...   ("" = nothing set); prints the top kernels of each.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
mkdir -p $REPO/gpurun_out/r4; cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  tag=$(echo "${kv:-default}" | tr -c 'A-Za-z0-9_\n' '_'); out=$REPO/gpurun_out/r4/abe_$tag; rm -rf $out
  ( [ -n "$kv" ] && export "$kv"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $REPO/$ARGS > $out.log 2>&1 )
  echo "== ${kv:-default}"; f=$(find $out -name "*kernel_stats.csv" | head -1)
  grep '^{' $out.log | python3 -c "import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  bench', d.get('config',{}).get('workload'), d.get('ms_per_step'), 'ms/step')" 2>/dev/null
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
