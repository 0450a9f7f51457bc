#!/bin/bash
# A/B library: one kernel file rebuilt with extra defines, linked with the other (unchanged) objects
#   [VARIANT_SRC=hidden_gather] tools/build_variant.sh <tag> -DFOO [-DBAR ...]   ->  variants/libgsage_<tag>.so
# (VARIANT_SRC defaults to project; run the variants with tools/variant_ab.sh)
tag=$1; shift
src=${VARIANT_SRC:-project}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/variants $root/build/variants
make -C $root/graphsage_amd/csrc -j8 > /dev/null || exit 1
extra=""; { [ $src = hidden_gather ] || [ $src = hidden_wgrad ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$root/include -Wall $extra "$@" \
  -c $root/graphsage_amd/csrc/$src.hip -o $root/build/variants/${src}_$tag.o || exit 1
objs=$(ls $root/build/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $root/build/variants/${src}_$tag.o -o $root/variants/libgsage_$tag.so
