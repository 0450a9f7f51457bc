#!/bin/bash
# A/B library: project.hip rebuilt with extra defines, linked with the other (unchanged) objects
#   tools/build_variant.sh <tag> -DFOO [-DBAR ...]   ->  variants/libgsage_<tag>.so   (tools/variant_ab.sh)
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/variants $root/build/variants
make -C $root/graphsage_amd/csrc -j8 > /dev/null || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$root/include -Wall "$@" \
  -c $root/graphsage_amd/csrc/project.hip -o $root/build/variants/project_$tag.o || exit 1
objs=$(ls $root/build/csrc/*.o | grep -v "/project")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $root/build/variants/project_$tag.o -o $root/variants/libgsage_$tag.so
