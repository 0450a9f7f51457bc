#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call (run from the repo root on the box):
#   tools/final_profiles.sh <round tag, e.g. r02>
# -> gpurun_out/<tag>_{c4,c5,c3}_kernel_stats.csv, <tag>_{c4,c5}_pmc_hbm.json (FETCH_SIZE, WRITE_SIZE
#    and RDREQ-size passes, each a run of its own, stamped with the kernel-source hash),
#    <tag>_bench_{c4,c5}.json.  Copy what is to be judged into profiles/ afterwards.
tag=$1
set -e
for w in c4 c5; do
  extra=""; [ $w = c4 ] && extra="--no-reference-batch"
  bash tools/kstats.sh ${tag}_$w --workload $w --steps 10 --warmup 3 $extra
  cp gpurun_out/kstats_${tag}_$w.csv gpurun_out/${tag}_${w}_kernel_stats.csv
  bash tools/pmc_pass.sh FETCH_SIZE --workload $w $extra
  bash tools/pmc_pass.sh WRITE_SIZE --workload $w $extra
  bash tools/pmc_multi.sh rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" --workload $w $extra
  python3 tools/pmc_hbm_merge.py gpurun_out/${tag}_${w}_pmc_hbm.json "bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1"
  echo "== $w profiled"
done
bash tools/kstats.sh ${tag}_c3 --workload c3 --steps 10 --warmup 3 --no-reference-batch
cp gpurun_out/kstats_${tag}_c3.csv gpurun_out/${tag}_c3_kernel_stats.csv
cp gpurun_out/${tag}_c4_pmc_hbm.json gpurun_out/${tag}_c5_pmc_hbm.json profiles/   # so that the bench lines below read this build's traffic
timeout -k 10 600 python3 bench.py > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.log
timeout -k 10 600 python3 bench.py --workload c5 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.log
echo "== done"
