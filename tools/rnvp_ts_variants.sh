#!/bin/bash
# Timing-only builds of the shared-hand-over B-ts kernel: tools/bin/libmnf_ts_<tag>.so for each "tag:flags" argument.
# usage (here): tools/rnvp_ts_variants.sh build abl1:-DMNF_RNVP_TS_ABL=1 ...   (GPU box): tools/rnvp_ts_variants.sh run abl1 ...
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/torch_mnf_amd/csrc
mode=$1; shift
mkdir -p $REPO/tools/bin
for arg in "$@"; do
  tag=${arg%%:*}; flags=${arg#*:}; [ "$flags" = "$arg" ] && flags=""
  so=$REPO/tools/bin/libmnf_ts_$tag.so
  if [ "$mode" = build ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -I$CS $flags -c $CS/mnf_rnvp_bwd.hip -o /tmp/ts_$tag.o || exit 1
    objs=$(ls $CS/*.o | grep -v mnf_rnvp_bwd.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $so $objs /tmp/ts_$tag.o || exit 1
  else
    cd /tmp && export TMPDIR=/tmp
    out=$REPO/gpurun_out/r4/ts_$tag; rm -rf $out
    MNF_LIB_PATH=$so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $REPO/tools/time_rnvp_bwd_only.py ${ROWS:-256000} 10 > /dev/null 2>&1
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "== $tag: $(grep -h "rnvp_bwd_ts" "$f" | awk -F'","|",|,"' '{print $1}' | cut -c1-50) avg_ns $(grep -h "rnvp_bwd_ts" "$f" | awk -F, '{print $(NF-4)}')"
  fi
done
