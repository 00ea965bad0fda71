#!/bin/bash
# GPU box: parity tests of the RNVP gradient kernels, then a same-box A/B of launch B-ts (MNF_RNVP_BWD_TS=split = the
# round-3 one-slab-per-workgroup kernel; default = the shared-hand-over kernel).  Writes gpurun_out/r4/ab_rnvp_bwd.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r4; mkdir -p $OUT
cd $REPO
{
echo "== tests"
timeout 900 python3 -m pytest tests/test_hip_round3.py -q -m gpu -x -k "rnvp or mnf_linear" 2>&1 | tail -5
for v in split shared; do
  echo "== MNF_RNVP_BWD_TS=$v"
  MNF_RNVP_BWD_TS=$v timeout 300 python3 tools/time_rnvp_bwd_only.py 256000 10
  MNF_RNVP_BWD_TS=$v timeout 300 python3 bench.py --workload c5t --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5t ms_per_step', d['ms_per_step'], 'kernel_us', d['roofline'].get('avg_kernel_us'))"
done
cd /tmp && export TMPDIR=/tmp
for v in split shared; do
  MNF_RNVP_BWD_TS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $REPO/tools/time_rnvp_bwd_only.py 256000 10 > /dev/null 2>&1
  echo "== kernel stats $v"; f=$(find $OUT/trace_$v -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-150
done
} > $OUT/ab_rnvp_bwd.txt 2>&1
cat $OUT/ab_rnvp_bwd.txt
