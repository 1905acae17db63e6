#!/bin/bash
# A/B of two builds of the library on the same GPU box: tools/ab_perf.sh a.so b.so [strain|demo|short ...]   (alternates a b a b)
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for h in ${@:-strain demo}; do
	for lib in $A $B $A $B; do
		if [ $h = short ]; then
			DSB_LIB_PATH=$PWD/$lib python bench.py --headline demo --steps 1 --warmup 0 --demo-steps 1 --demo-batches 1 --no-seed-hbm --no-cli --no-cpu-baseline --no-end-to-end --no-proxy > gpurun_out/ab.json 2> gpurun_out/ab.err
			python -c "
import json; d=json.load(open('gpurun_out/ab.json'))['config2_short_reads']; print('$h $lib: %.2f M reads/s  k_classify %.1f ms' % (d['reads_per_s'] / 1e6, d['kernel_ms']['k_classify']))"
		else
			DSB_LIB_PATH=$PWD/$lib python bench.py --headline $h --steps 4 --warmup 1 --no-demo-index --no-cli --no-cpu-baseline --no-end-to-end --no-short-reads --no-proxy --no-seed-hbm --no-budget-build > gpurun_out/ab.json 2> gpurun_out/ab.err
			python -c "
import json; d=json.load(open('gpurun_out/ab.json')); k=d['kernel_ms_per_step']; print('$h $lib: %.0f reads/s  k_classify %.1f ms  seed %.1f ms' % (d['value'], k['k_classify'], k.get('k_seed_scan', 0)))"
		fi
	done
done
