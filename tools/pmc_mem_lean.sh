#!/bin/bash
# Memory-pipe counters of one bench workload, few counters per pass (a pass that asks for more TCC / TA counters than the
# hardware has slots for aborts: "Request exceeds the capabilities of the hardware to collect").
# usage: tools/pmc_mem_lean.sh <tag> [bench args...]    -> gpurun_out/prof_<tag>/mem_lean.txt
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --prime-ms 10 $*"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $set --output-format csv -d "$OUT/lean$i" -- python3 "$REPO/bench.py" $ARGS > "$OUT/lean$i.log" 2>&1 || echo "pass $i ($set) failed"
done
python3 - "$OUT" <<'PY' | tee "$OUT/mem_lean.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/lean*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("rnvp", "ahf", "nsf", "mnf_linear")): continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:44s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
