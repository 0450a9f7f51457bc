"""Per-call durations of kernels whose name contains a pattern, from a rocprofv3 kernel_trace.csv:
python tools/ktrace_calls.py trace.csv pattern [pattern ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    print(pat, len(d), "calls (us):", " ".join(f"{x:.0f}" for x in d[:40]))
