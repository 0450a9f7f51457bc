"""Instruction mix of the basic blocks that hold MFMAs in one kernel of a device assembly file
(hipcc --cuda-device-only -S): python tools/asm_loop_count.py file.s <substring of the mangled name>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
    else:
        cur.append(l.strip())
blocks.append((name, cur))
tot = Counter()
for name, b in blocks:
    c = Counter()
    for l in b:
        if not l or l.startswith(";") or l.startswith("."):
            continue
        o = l.split()[0]
        k = ("mfma" if o.startswith("v_mfma") else "valu" if o.startswith("v_") else "waitcnt" if o.startswith("s_waitcnt")
             else "salu" if o.startswith("s_") else "vmem" if o.startswith(("global_", "buffer_", "flat_")) else
             "lds" if o.startswith("ds_") else "other")
        c[k] += 1
    tot.update(c)
    if c["mfma"]:
        print(name, dict(c))
print("whole kernel", dict(tot))
