#!/bin/bash
# GPU box: same-box A/B of library builds on one script.  usage: tools/ab_libs.sh "<script args>" lib1.so lib2.so ...
# ("" = the in-tree library); prints the kernel stats (top 8) of each.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
mkdir -p $REPO/gpurun_out/r4; cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "${lib:-intree}" .so); out=$REPO/gpurun_out/r4/ab_$tag; rm -rf $out
  if [ -n "$lib" ]; then export MNF_LIB_PATH=$REPO/$lib; else unset MNF_LIB_PATH; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $REPO/$ARGS > $out.log 2>&1
  echo "== $tag"; f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
