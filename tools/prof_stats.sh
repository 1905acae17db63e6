#!/bin/bash
# rocprofv3 kernel-trace statistics of the bench command (one batch, 4 steps): writes gpurun_out/<tag>_kernel_stats.csv
#   tools/prof_stats.sh <tag> [reads]        PROF_HEADLINE=demo: the demo index instead of the strain index built in the bench; PROF_SEED_HBM_MIB=<MiB>: tools/seed_hbm_only.py
set -e -o pipefail
export DSB_NO_WARMUP=1      # (the four-read batch dsb_ctx_create runs with hints would count as a launch of every kernel)
cd "$(dirname "$0")/.."
ROOT=$PWD; tag=$1; reads=${2:-65536}
export TMPDIR=/tmp
out=$ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$ROOT/gpurun_out"
if [ -n "$PROF_SEED_HBM_MIB" ]; then (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 "$ROOT/tools/seed_hbm_only.py" "$reads" "$PROF_SEED_HBM_MIB" > "$out.json" 2> "$out.err"); else
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 "$ROOT/bench.py" --headline "${PROF_HEADLINE:-strain}" --no-cpu-baseline --no-end-to-end --no-cli --no-demo-index --no-proxy --no-short-reads --no-budget-build --steps 4 --warmup 1 --batches 2 --reads-per-gpu "$reads" > "$out.json" 2> "$out.err"); fi
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/${tag}_kernel_stats.csv"
cp "$out.json" "$ROOT/gpurun_out/${tag}_bench_under_rocprof.json"
cat "$ROOT/gpurun_out/${tag}_kernel_stats.csv"
