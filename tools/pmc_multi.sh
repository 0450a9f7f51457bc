#!/bin/bash
# several counters in one run:  tools/pmc_multi.sh <tag> "<CTR1 CTR2 ...>" [bench args]
# (PMC_SCRIPT=tools/x.py profiles that script instead of bench.py)
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm_$tag
if [ -n "$PMC_SCRIPT" ]; then
  timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcm_$tag -o r -- python3 /root/repo/$PMC_SCRIPT "$@" > /root/repo/gpurun_out/pmcm_$tag.log 2>&1 || exit 1
else
  timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcm_$tag -o r -- python3 /root/repo/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /root/repo/gpurun_out/pmcm_$tag.log 2>&1 || exit 1
fi
f=$(find /tmp/pmcm_$tag -name '*counter_collection.csv' | head -1)
python3 - "$f" > /root/repo/gpurun_out/pmcm_$tag.txt <<'PY'
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    m = re.match(r"(?:void )?\(anonymous namespace\)::(\w+(?:<[^>]*>)?)\(", row["Kernel_Name"])
    if m:
        acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
