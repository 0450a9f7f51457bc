#!/bin/bash
# run a probe script once per prebuilt library variant: tools/variant_probe.sh "<tags>" tools/x.py [args]
tags=$1; shift
cp graphsage_amd/lib/libgsage.so /tmp/orig.so
for v in $tags; do
  if [ $v = orig ]; then cp /tmp/orig.so graphsage_amd/lib/libgsage.so; else cp variants/libgsage_$v.so graphsage_amd/lib/libgsage.so; fi
  echo "== $v"; timeout -k 10 300 python3 "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
cp /tmp/orig.so graphsage_amd/lib/libgsage.so
