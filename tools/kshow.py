"""Short view of a rocprofv3 kernel_stats.csv: python tools/kshow.py file.csv [min_total_ms]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
for r in rows:
    tot = float(r["TotalDurationNs"]) / 1e6
    if tot < lim:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    print(f"{name:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e6:9.4f} ms  total {tot:9.2f} ms")
