#!/bin/bash
# copies what tools/final_profiles.sh <tag> (+ tools/ec_probe.sh, + a full `python bench.py > gpurun_out/<tag>_final_bench_512.json`) left under
# gpurun_out/ into profiles/<tag>_final_*  (run here, after the gpurun calls)
set -e
T=${1:-r5}
cp gpurun_out/prof_${T}_ng/stats/ks_kernel_stats.csv profiles/${T}_final_kernel_stats_512_neargrid.csv
cp gpurun_out/prof_${T}_og/stats/ks_kernel_stats.csv profiles/${T}_final_kernel_stats_512_ongrid.csv
cp gpurun_out/prof_${T}_ng/pmc_fetch_write.txt profiles/${T}_final_pmc_fetch_write_512_neargrid.txt
cp gpurun_out/prof_${T}_ng/pmc_fetch_write_steady.txt profiles/${T}_final_pmc_fetch_write_steady_512_neargrid.txt
cp gpurun_out/prof_${T}_og/pmc_fetch_write.txt profiles/${T}_final_pmc_fetch_write_512_ongrid.txt
cp gpurun_out/prof_${T}_og/pmc_fetch_write_steady.txt profiles/${T}_final_pmc_fetch_write_steady_512_ongrid.txt
cp gpurun_out/final_${T}/pmc_sq_512_neargrid.txt profiles/${T}_final_pmc_sq_512_neargrid.txt
cp gpurun_out/final_${T}/stats_1024/ks_kernel_stats.csv profiles/${T}_final_kernel_stats_1024_neargrid.csv
cp gpurun_out/final_${T}/stats_8slabs/ks_kernel_stats.csv profiles/${T}_final_kernel_stats_512_8slabs_emulated.csv
cp gpurun_out/final_${T}/slab_emulation_512.jsonl profiles/${T}_final_slab_emulation_512.jsonl
cp gpurun_out/final_${T}/slab_emulation_1024.jsonl profiles/${T}_final_slab_emulation_1024.jsonl
for s in 64 256 1024; do cp gpurun_out/final_${T}/bench_$s.json profiles/${T}_final_bench_$s.json; done
cp gpurun_out/prof_${T}_og/bench.json profiles/${T}_final_bench_512_ongrid.json
[ -f gpurun_out/ec_probe.txt ] && cp gpurun_out/ec_probe.txt profiles/${T}_final_ec_probe_512.txt
[ -f gpurun_out/${T}_final_bench_512.json ] && cp gpurun_out/${T}_final_bench_512.json profiles/${T}_final_bench_512.json
head -1 profiles/${T}_final_pmc_fetch_write_steady_512_neargrid.txt
