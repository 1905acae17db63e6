#!/bin/bash
# A/B of two builds of the library on the same GPU box: tools/ab_perf.sh a.so b.so [strain|demo ...]   (alternates a b a b)
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for h in ${@:-strain demo}; do
	for lib in $A $B $A $B; do
		DSB_LIB_PATH=$PWD/$lib python bench.py --headline $h --steps 4 --warmup 1 --no-demo-index --no-cli --no-cpu-baseline --no-end-to-end --no-short-reads > gpurun_out/ab.json 2> gpurun_out/ab.err
		python -c "
import json; d=json.load(open('gpurun_out/ab.json')); k=d['kernel_ms_per_step']; print('$h $lib: %.0f reads/s  k_classify %.1f ms' % (d['value'], k['k_classify']))"
	done
done
