"""Mean counter value per launch and kernel from a rocprofv3 --pmc csv (values as reported)."""
import csv
import json
import re
import sys
from collections import defaultdict

path, ctr = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
with open(path) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != ctr:
            continue
        name = row["Kernel_Name"]
        m = re.match(r"(?:void )?\(anonymous namespace\)::(\w+(?:<[^>]*>)?)\(", name)
        if not m:
            continue                      # not one of libgsage's kernels
        acc[m.group(1)].append(float(row["Counter_Value"]))
print(json.dumps({k: {"launches": len(v), "mean_KB": sum(v) / len(v)} for k, v in sorted(acc.items())},
                 indent=1))
