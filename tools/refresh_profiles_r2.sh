#!/bin/bash
# Regenerates the round-2 evidence on the GPU box into gpurun_out/r2/ (copy what is to be judged into profiles/r2/):
# bench JSON lines, rocprofv3 kernel-trace + PMC summaries (every profiler run under `timeout`), PMC traffic files.
# usage: tools/refresh_profiles_r2.sh [workloads...]   (default: c2 c3 c4 c5; c2t = the training step)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2
mkdir -p "$OUT"
cd "$REPO"
WL=${*:-c2 c3 c4 c5}
declare -A KERN=( [c2]="ahf_split_stack_kernel<32, 24, true" [c4]="ahf_split_stack_kernel<128, 24, true" \
                  [c3]="nsf_mfma_kernel<16, 8, 8, true, 2, true" [c5]="rnvp_resident_kernel<50, 50, false" \
                  [c2t]="ahf_bwd_split_kernel<32, 24, true" )
for w in $WL; do
  python bench.py --workload $w > "$OUT/${w}_bench.json" 2> "$OUT/${w}_bench.err"
  tools/profile_bench.sh r2_$w --workload $w > /dev/null 2>&1
  cp "$REPO/gpurun_out/prof_r2_$w/summary.txt" "$OUT/${w}_rocprofv3_summary.txt"
  cp "$REPO"/gpurun_out/prof_r2_$w/trace/*/*kernel_stats.csv "$OUT/${w}_kernel_stats.csv" 2>/dev/null
  python tools/make_traffic_json.py gpurun_out/prof_r2_$w $w "${KERN[$w]}" "$OUT/${w}_pmc_traffic.json" > /dev/null 2>&1
  python tools/make_valu_json.py gpurun_out/prof_r2_$w $w "${KERN[$w]}" "$OUT/${w}_valu_issue.json" > /dev/null 2>&1
done
ls -la "$OUT"
for f in "$OUT"/*_bench.json; do echo "$f"; tail -1 "$f" | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('  ', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', r['bound'], round(r['frac'],3), round(r['avg_kernel_us'],1), 'us', r.get('traffic'))"; done
for f in "$OUT"/*_pmc_traffic.json; do echo "$f"; grep -E "FETCH|WRITE|traffic_bytes" "$f"; done
