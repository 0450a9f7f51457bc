"""Merge the per-counter summaries of tools/pmc_pass.sh (gpurun_out/pmc_FETCH_SIZE.json,
pmc_WRITE_SIZE.json, optional pmcm_rdreq.txt) into one profiles/*_pmc_hbm.json and stamp it with
the hash of the kernel sources the profiled library was built from (bench.py: csrc_hash), so
bench.py can tell when roofline.traffic no longer belongs to the build it is timing.

  python tools/pmc_hbm_merge.py <out.json> [note]
"""
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = {"csrc_hash": bench.csrc_hash(), "note": sys.argv[2] if len(sys.argv) > 2 else "",
       "units": "mean_KB = counter value per launch as rocprofv3 reports it (KiB); FETCH_SIZE counts "
                "every fabric read request as 64 B while the RDREQ size counters show all of them "
                "to be 128-B requests on gfx950, hence bytes = 2 * FETCH_SIZE * 1024"}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    z = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc_{ctr}.json")))
    out[ctr] = {k: v for k, v in z.items() if "at::" not in k and "rocprim" not in k}
rd = os.path.join(ROOT, "gpurun_out", "pmcm_rdreq.txt")
if os.path.exists(rd):
    sizes = {}
    for line in open(rd):
        name, _, rest = line.partition(" {")
        if "at::" in name or "rocprim" in name or not rest:
            continue
        d = ast.literal_eval("{" + rest.strip())
        sizes[name] = {"requests": d.get("TCC_EA0_RDREQ_sum"), "of_32B": d.get("TCC_EA0_RDREQ_32B_sum"),
                       "of_64B": d.get("TCC_EA0_RDREQ_64B_sum"), "of_128B": d.get("TCC_EA0_RDREQ_128B_sum"),
                       "bytes": 32 * d.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * d.get("TCC_EA0_RDREQ_64B_sum", 0)
                       + 128 * d.get("TCC_EA0_RDREQ_128B_sum", 0)}
    out["RDREQ_SIZES"] = sizes
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], "csrc_hash", out["csrc_hash"])
