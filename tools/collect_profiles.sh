#!/bin/bash
# Copies what a GPU run of tools/r03_profiles.sh + bench.py left under gpurun_out/ to the names profiles/ is judged by.
#   tools/collect_profiles.sh <bench-json-in-gpurun_out>
cd "$(dirname "$0")/.."
cp gpurun_out/pmc_r03_strain.json profiles/r03_pmc_traffic_strain.json
cp gpurun_out/pmc_r03_demo.json profiles/r03_pmc_traffic_demo.json
cp gpurun_out/pmc_r03_seed_hbm_2g.json profiles/r03_pmc_seed_hbm.json
cp gpurun_out/pmc_r03_seed_hbm_8g.json profiles/r03_pmc_seed_hbm_8g.json
cp gpurun_out/r03_strain_kernel_stats.csv gpurun_out/r03_demo_kernel_stats.csv profiles/
cp gpurun_out/r03_seed_hbm_2g_kernel_stats.csv profiles/r03_seed_hbm_2x2GiB_kernel_stats.csv
cp gpurun_out/r03_seed_hbm_8g_kernel_stats.csv profiles/r03_seed_hbm_2x8GiB_kernel_stats.csv
cp gpurun_out/r03_strain_bench_under_rocprof.json gpurun_out/r03_demo_bench_under_rocprof.json profiles/
tail -1 "gpurun_out/$1" > profiles/r03_bench.json
cp gpurun_out/gpu_tests.log profiles/r03_gpu_tests.log
