#!/bin/bash
# per-kernel averages of a short bench.py run: tools/kstats.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o r -- python3 /root/repo/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > /root/repo/gpurun_out/kstats_$tag.log 2>&1 || exit 1
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" /root/repo/gpurun_out/kstats_$tag.csv
