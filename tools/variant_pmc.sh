#!/bin/bash
# PMC pass of a probe script per prebuilt library variant:
#   tools/variant_pmc.sh "<tags>" "<counters>" <kernel regex> tools/x.py [args]
tags=$1; ctrs=$2; kre=$3; shift 3
script=$1; shift
cp graphsage_amd/lib/libgsage.so /tmp/orig.so
for v in $tags; do
  if [ $v = orig ]; then cp /tmp/orig.so graphsage_amd/lib/libgsage.so; else cp variants/libgsage_$v.so graphsage_amd/lib/libgsage.so; fi
  PMC_SCRIPT=$script bash tools/pmc_multi.sh vp_$v "$ctrs" "$@" || { echo "pmc pass failed for $v"; cp /tmp/orig.so graphsage_amd/lib/libgsage.so; exit 1; }
  echo "== $v"; grep -E "$kre" gpurun_out/pmcm_vp_$v.txt
done
cp /tmp/orig.so graphsage_amd/lib/libgsage.so
