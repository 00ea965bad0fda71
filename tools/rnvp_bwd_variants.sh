#!/bin/bash
# Ablation builds of the RNVP gradient kernels (timing only): tools/bin/libmnf_bwd_abl<N>.so, one per MNF_RNVP_BWD_ABL
# value given on the command line (see mnf_rnvp_bwd.hip), each timed with tools/time_rnvp_bwd.py on the GPU box.
# usage (here): tools/rnvp_bwd_variants.sh build 0 1 2 4 8 ...   (GPU box): tools/rnvp_bwd_variants.sh run 0 1 2 4 8 ...
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/torch_mnf_amd/csrc
mode=$1; shift
mkdir -p $REPO/tools/bin
for v in "$@"; do
  so=$REPO/tools/bin/libmnf_bwd_abl$v.so
  if [ "$mode" = build ]; then
    name=abl${v//[^A-Za-z0-9]/_}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics ${EXTRA:-} -DMNF_RNVP_BWD_ABL=$v -c $CS/mnf_rnvp_bwd.hip -o /tmp/$name.o || exit 1
    objs=$(ls $CS/*.o | grep -v mnf_rnvp_bwd.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $so $objs /tmp/$name.o || exit 1
  else
    echo "== ABL $v: $(MNF_LIB_PATH=$so python3 $REPO/tools/time_rnvp_bwd.py ${ROWS:-256000} 2>&1 | grep rows)"
    MNF_LIB_PATH=$so python3 - <<PY
import os, sys
sys.path.insert(0, "$REPO")
import torch, torch_mnf_amd as amd
rows = int(os.environ.get("ROWS", "256000"))
f = amd.RNVP(800, h_sizes=(50,)).to("cuda")
z = torch.randn(rows, 800, device="cuda", requires_grad=True)
w = torch.randn(rows, 800, device="cuda") / rows
x, ld = f.forward(z, seed=7)
gx = w; gl = torch.full((rows,), 1.0 / rows, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(6):
    if i == 1: ev[0].record()
    torch.autograd.grad((x, ld), (z, *f.parameters()), (gx, gl), retain_graph=True)
ev[1].record(); torch.cuda.synchronize()
print("   backward only: %.3f ms" % (ev[0].elapsed_time(ev[1]) / 5))
PY
  fi
done
