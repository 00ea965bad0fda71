#!/bin/bash
# Regenerates a round's evidence on the GPU box into gpurun_out/$ROUND/ (copy what is to be judged into profiles/$ROUND/):
# bench JSON lines, rocprofv3 kernel-trace + PMC summaries (every profiler run under `timeout`), PMC traffic files.
# usage: [ROUND=r4] tools/refresh_profiles.sh [workloads...]   (default: c2 c3 c4 c5 c5b c6 c2t c3t c5t)
set -u
ROUND=${ROUND:-r5}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$ROUND
mkdir -p "$OUT"
cd "$REPO"
WL=${*:-c2 c3 c4 c5 c5b c6 c2t c3t c5t}
declare -A KERN=( [c2]="ahf_split_stack_kernel<32, 24, true" [c4]="ahf_split_stack_kernel<128, 24, true" \
                  [c3]="nsf_mfma_kernel<16, 8, 8, true, 2, true" [c5]="rnvp_resident_kernel<50, 50, false" \
                  [c2t]="ahf_bwd_split_kernel<32, 24, true" [c5t]="rnvp_bwd_ts_shared_kernel<50, true, false" \
                  [c3t]="nsf_bwd_tile_kernel" [c5b]="rnvp_narrow_kernel<50" [c6]="ahf_rt_kernel<4, 1, 8, true, 1>" )
declare -A MULT=( [c3t]=2 )
for w in $WL; do
  extra=""; [ "$w" = c2 ] && extra="--no-secondary"
  timeout 400 python bench.py --workload $w $extra > "$OUT/${w}_bench.json" 2> "$OUT/${w}_bench.err"
  P=$REPO/gpurun_out/prof_${ROUND}_$w
  mkdir -p "$P"
  ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --workload $w"
  [ "$w" = c5t ] && ARGS="--steps 6 --warmup 2 --no-cpu-baseline --workload $w"
  [ "$w" = c3t ] && ARGS="--steps 6 --warmup 2 --no-cpu-baseline --workload $w"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 "$REPO/bench.py" $ARGS > "$P/trace.log" 2>&1
    timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$P/pmc_fetch.log" 2>&1
    timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$P/pmc_write.log" 2>&1
    if true; then
      timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$P/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$P/pmc_sq.log" 2>&1
      timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$P/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > "$P/pmc_sq2.log" 2>&1
    fi )
  python3 tools/summarize_prof.py "$P" > "$OUT/${w}_rocprofv3_summary.txt" 2>&1
  cp "$P"/trace/*/*kernel_stats.csv "$OUT/${w}_kernel_stats.csv" 2>/dev/null || cp "$P"/trace/*kernel_stats.csv "$OUT/${w}_kernel_stats.csv" 2>/dev/null
  python3 tools/make_traffic_json.py gpurun_out/prof_${ROUND}_$w $w "${KERN[$w]}" "$OUT/${w}_pmc_traffic.json" ${MULT[$w]:-1} > /dev/null 2>&1
  python3 tools/make_valu_json.py gpurun_out/prof_${ROUND}_$w $w "${KERN[$w]}" "$OUT/${w}_valu_issue.json" ${MULT[$w]:-1} > /dev/null 2>&1
done
ls -la "$OUT"
for f in "$OUT"/*_bench.json; do echo "$f"; tail -1 "$f" | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('  ', d['value'], d['unit'], d['ms_per_step'], 'ms', r['bound'], r['frac'], r['avg_kernel_us'], 'us', r.get('traffic'))"; done
for f in "$OUT"/*_pmc_traffic.json; do echo "$f"; grep -E "FETCH|WRITE|traffic_bytes" "$f"; done
