#!/bin/bash
# A variant of the library with one source recompiled under extra flags (A/B experiments on one box):
#   tools/build_variant.sh <name> <source.hip> [flags...]  ->  dreammat_amd/csrc/_obj/<name>/libdreammat_hip.so
# Use it with DREAMMAT_LIB=<that path> (python) or LD_LIBRARY_PATH=<that dir> (tools/_abi_pmc).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
O=$R/dreammat_amd/csrc/_obj
mkdir -p $O/$name
extra=""
case $src in attention.hip|attn_w64.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; shade.hip) extra="-munsafe-fp-atomics -ffast-math";; hashgrid.hip) extra="-munsafe-fp-atomics";; raster.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $extra "$@" -c $R/dreammat_amd/csrc/$src -o $O/$name/$src.o
objs=""
for o in $O/*.o; do b=$(basename $o); if [ "$b" = "$src.o" ]; then objs="$objs $O/$name/$src.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/$name/libdreammat_hip.so $objs
echo $O/$name/libdreammat_hip.so
