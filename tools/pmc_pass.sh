#!/bin/bash
# one PMC counter, one run (rocprofv3 does not split counters over passes):
#   tools/pmc_pass.sh <COUNTER> [bench args...]   ->  gpurun_out/pmc_<COUNTER>.csv
ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$ctr
timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o r -- python3 /root/repo/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /root/repo/gpurun_out/pmc_$ctr.log 2>&1 || exit 1
f=$(find /tmp/pmc_$ctr -name '*counter_collection.csv' | head -1)
python3 /root/repo/tools/pmc_summarise.py "$f" $ctr > /root/repo/gpurun_out/pmc_$ctr.json
