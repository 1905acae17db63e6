#!/bin/bash
# rocprofv3 kernel-trace statistics of the bench command (one batch, 4 steps): writes gpurun_out/<tag>_kernel_stats.csv
#   tools/prof_stats.sh <tag> [reads]
set -e -o pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD; tag=$1; reads=${2:-65536}
export TMPDIR=/tmp
out=$ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$ROOT/gpurun_out"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-end-to-end --no-seed-hbm --no-short-reads --no-second-index --steps 4 --warmup 1 --batches 2 --reads-per-gpu "$reads" > "$out.json" 2> "$out.err")
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/${tag}_kernel_stats.csv"
cp "$out.json" "$ROOT/gpurun_out/${tag}_bench_under_rocprof.json"
cat "$ROOT/gpurun_out/${tag}_kernel_stats.csv"
