#!/bin/bash
# Regenerates everything under profiles/r1/ on the GPU box (writes to gpurun_out/r1/, copy from there):
#   bench JSON lines for every workload (default path, plus the A/B switches), rocprofv3 kernel-trace + PMC
#   summaries for c2 / c3 / c5, the training-step timings.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r1
mkdir -p "$OUT" "$OUT/fp32_mfma"
cd "$REPO"
python bench.py > "$OUT/c2_bench.json" 2> "$OUT/c2_bench.err"
for w in c4 c3 c5; do python bench.py --workload $w > "$OUT/${w}_bench.json" 2> "$OUT/${w}_bench.err"; done
python bench.py --workload c2f --no-cpu-baseline > "$OUT/c2_fused_stack_optin_bench.json" 2>/dev/null
python bench.py --workload c3f --no-cpu-baseline > "$OUT/c3_fused_optin_bench.json" 2>/dev/null
MNF_NO_RUN_FUSION=1 python bench.py --no-cpu-baseline > "$OUT/c2_layer_by_layer_bench.json" 2>/dev/null
MNF_NO_RUN_FUSION=1 python bench.py --workload c3 --no-cpu-baseline > "$OUT/c3_layer_by_layer_bench.json" 2>/dev/null
MNF_NO_RUN_FUSION=1 python bench.py --workload c4 --no-cpu-baseline > "$OUT/c4_layer_by_layer_bench.json" 2>/dev/null
MNF_FP32_MFMA=1 python bench.py --no-cpu-baseline > "$OUT/fp32_mfma/c2_bench_same_build.json" 2>/dev/null
MNF_FP32_MFMA=1 MNF_NO_RUN_FUSION=1 python bench.py --no-cpu-baseline > "$OUT/fp32_mfma/c2_layer_by_layer_bench_same_build.json" 2>/dev/null
MNF_FP32_MFMA=1 python bench.py --workload c5 --no-cpu-baseline > "$OUT/fp32_mfma/c5_bench_same_build.json" 2>/dev/null
python tools/bench_backward.py > "$OUT/training_step.txt" 2>&1
python tools/bench_c1.py > "$OUT/c1_latency.txt" 2>&1
python tools/bench_c5.py 2>/dev/null | grep MNFLinear > "$OUT/c5_sample_z.txt"
python tools/mnf_lenet_harness.py 2>/dev/null | grep MNF-LeNet > "$OUT/c5_mnf_lenet_forward.txt"
for w in c2 c3 c5; do
  tools/profile_bench.sh r1_$w --workload $w > /dev/null 2>&1
  cp "$REPO/gpurun_out/prof_r1_$w/summary.txt" "$OUT/${w}_rocprofv3_summary.txt"
  cp "$REPO"/gpurun_out/prof_r1_$w/trace/*/*kernel_stats.csv "$OUT/${w}_kernel_stats.csv"
done
python tools/make_traffic_json.py gpurun_out/prof_r1_c2 c2 "ahf_split_stack_kernel<32, 24, true" "$OUT/c2_pmc_traffic.json" > /dev/null
python tools/make_traffic_json.py gpurun_out/prof_r1_c5 c5 "rnvp_split_kernel<50, true" "$OUT/c5_pmc_traffic.json" > /dev/null
python tools/make_traffic_json.py gpurun_out/prof_r1_c3 c3 "nsf_mfma_kernel<16, 8, 8, true, 2, true" "$OUT/c3_pmc_traffic.json" > /dev/null
ls -la "$OUT"
for f in "$OUT"/*_bench.json; do echo "$f"; tail -1 "$f" | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('  ', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', r['bound'], round(r['frac'],3), round(r['avg_kernel_us'],1), 'us', r.get('traffic'))"; done
