#!/bin/bash
# per-kernel averages of an arbitrary python command: tools/kstats_cmd.sh <tag> script.py [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
script=/root/repo/$1; shift
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o r -- python3 $script "$@" > /root/repo/gpurun_out/kstats_$tag.log 2>&1 || exit 1
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" /root/repo/gpurun_out/kstats_$tag.csv
