#!/bin/bash
# Copies what a GPU run of tools/r04_profiles.sh + bench.py left under gpurun_out/ to the names profiles/ is judged by.
#   tools/collect_profiles.sh <bench-json-in-gpurun_out>
cd "$(dirname "$0")/.."
cp gpurun_out/pmc_r04_strain.json profiles/r04_pmc_traffic_strain.json
cp gpurun_out/pmc_r04_demo.json profiles/r04_pmc_traffic_demo.json
cp gpurun_out/pmc_r04_seed_hbm_2g.json profiles/r04_pmc_seed_hbm.json
cp gpurun_out/pmc_r04_seed_hbm_8g.json profiles/r04_pmc_seed_hbm_8g.json
cp gpurun_out/r04_strain_kernel_stats.csv gpurun_out/r04_demo_kernel_stats.csv profiles/
cp gpurun_out/r04_seed_hbm_2g_kernel_stats.csv profiles/r04_seed_hbm_2x2GiB_kernel_stats.csv
cp gpurun_out/r04_seed_hbm_8g_kernel_stats.csv profiles/r04_seed_hbm_2x8GiB_kernel_stats.csv
cp gpurun_out/r04_strain_bench_under_rocprof.json gpurun_out/r04_demo_bench_under_rocprof.json profiles/
tail -1 "gpurun_out/$1" > profiles/r04_bench.json
cp gpurun_out/gpu_tests.log profiles/r04_gpu_tests.log
