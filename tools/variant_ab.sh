#!/bin/bash
# A/B of prebuilt library variants on the GPU box: tools/variant_ab.sh "<tags>" <kernel regex> [bench args]
# (variants/libgsage_<tag>.so are copied over graphsage_amd/lib/libgsage.so in the box's scratch
# copy of the tree, one after the other; "orig" = the library as built)
tags=$1; kre=$2; shift 2
cp graphsage_amd/lib/libgsage.so /tmp/orig.so
for v in $tags; do
  if [ $v = orig ]; then cp /tmp/orig.so graphsage_amd/lib/libgsage.so; else cp variants/libgsage_$v.so graphsage_amd/lib/libgsage.so; fi
  bash tools/kstats.sh ab_$v "$@" || exit 1
  echo "== $v"; python3 tools/kshow.py gpurun_out/kstats_ab_$v.csv 0.0 | grep -E "$kre"
done
cp /tmp/orig.so graphsage_amd/lib/libgsage.so
