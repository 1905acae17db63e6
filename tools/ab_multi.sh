#!/bin/bash
# several builds of the library on one GPU box, two rounds: tools/ab_multi.sh strain|demo a.so b.so c.so ...
cd "$(dirname "$0")/.."
h=$1; shift
for rep in 1 2; do for lib in "$@"; do
	DSB_LIB_PATH=$PWD/$lib python bench.py --headline $h --steps 4 --warmup 1 --no-demo-index --no-cli --no-cpu-baseline --no-end-to-end --no-short-reads --no-proxy --no-seed-hbm --no-budget-build > gpurun_out/ab.json 2> gpurun_out/ab.err
	python -c "
import json; d=json.load(open('gpurun_out/ab.json')); k=d['kernel_ms_per_step']; print('$h $lib: %.0f reads/s  k_classify %.1f ms  seed %.1f ms' % (d['value'], k['k_classify'], k.get('k_seed_scan', 0)))"
done; done
