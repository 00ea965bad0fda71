#!/bin/bash
# Build tools/bin/libmnf_<tag>.so with ONE translation unit replaced by a modified source (same-box A/B builds).
# usage: tools/lib_variant.sh <tag> <object name, e.g. mnf_nsf_bwd_rows> <source file> [extra hipcc flags...]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd); CS=$REPO/torch_mnf_amd/csrc
tag=$1; obj=$2; src=$3; shift 3
mkdir -p $REPO/tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$CS -I$REPO/include "$@" -c $src -o /tmp/var_$tag.o || exit 1
objs=$(ls $CS/*.o | grep -v "/$obj.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/bin/libmnf_$tag.so $objs /tmp/var_$tag.o
