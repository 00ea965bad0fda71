#!/bin/bash
# Where do a kernel's cycles go: SQ counter passes over tools/time_rt.py (or another script).
# usage: tools/pmc_sq.sh <tag> [script args...]      env: MNF_LIB_PATH, MNF_NSF_BWD_KERNEL ... pass through; SCRIPT=tools/x.py
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCRIPT=${SCRIPT:-tools/time_rt.py}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p "$OUT"
case "$MNF_LIB_PATH" in ""|/*) ;; *) export MNF_LIB_PATH=$REPO/$MNF_LIB_PATH;; esac
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$REPO/$SCRIPT" "$@" > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("rnvp", "ahf", "nsf", "mnf_linear")): continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in acc.items():
        for line in [k] + [f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}" for c, v in sorted(cs.items())]:
            print(line); fh.write(line + "\n")
PY
