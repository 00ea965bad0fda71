#!/bin/bash
# A/B libraries of a register-resident RNVP kernel: recompiles mnf_rnvp_resident.hip (or, with SRC=mnf_rnvp_pair,
# the two-waves-per-tile kernel) with experiment switches and links it with the other objects into
# tools/bin/libmnf_<tag>.so (run here; the .so files travel to the GPU box).
# usage: [SRC=mnf_rnvp_pair] tools/kernel_variants.sh tag1:"-DMNF_RES_ABL=1" tag2:"-DMNF_RES_KC=2 -DMNF_RES_MC=2" ...
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/torch_mnf_amd/csrc
mkdir -p $REPO/tools/bin
SRC=${SRC:-mnf_rnvp_resident}
# per-file flags as in csrc/Makefile
case $SRC in
  mnf_rnvp_resident|mnf_rnvp_pair) BASE="-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-spill-vgpr-to-agpr=0";;
  mnf_ahf_bwd_split) BASE="-mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-spill-vgpr-to-agpr=0 -fno-honor-nans -fno-slp-vectorize";;
  mnf_ahf_*) BASE="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize";;
  *) BASE="-mllvm -amdgpu-mfma-vgpr-form";;
esac
OBJS=$(ls $C/*.o | grep -v $SRC.o)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $BASE $flags \
      -c $C/$SRC.hip -o /tmp/res_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/bin/libmnf_$tag.so $OBJS /tmp/res_$tag.o
  echo "built tools/bin/libmnf_$tag.so ($flags)"
done
