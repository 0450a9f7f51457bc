#!/bin/bash
# after tools/final_profiles.sh <tag> ran on the GPU box: copy what it left in gpurun_out/ into profiles/
tag=$1
for f in c3_kernel_stats.csv c4_kernel_stats.csv c5_kernel_stats.csv c4_pmc_hbm.json c5_pmc_hbm.json bench_c4.json bench_c5.json; do
  cp gpurun_out/${tag}_$f profiles/${tag}_$f || exit 1
done
python3 - "$tag" <<'PY'
import json, sys
for w in ("c4", "c5"):
    z = json.load(open(f"profiles/{sys.argv[1]}_bench_{w}.json"))
    r = z["roofline"]
    print(w, f'{z["ms_per_step"]:.2f} ms', f'{z["value"]:.4g} {z["unit"]}', r["bound"], f'frac {r["frac"]:.3f}',
          "traffic_stale", r.get("traffic_stale"))
PY
