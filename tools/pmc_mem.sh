#!/bin/bash
# Memory-path counter passes for one bench workload.  usage: tools/pmc_mem.sh <tag> [bench args...]
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --prime-ms 10 $*"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_mem$i" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_mem$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_mem*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rnvp" not in k and "ahf" not in k and "nsf" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:44s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
