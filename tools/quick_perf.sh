#!/bin/bash
# k_classify on both workloads, a few steps each (development loop): tools/quick_perf.sh [tag]
cd "$(dirname "$0")/.."
for h in strain demo; do
	python bench.py --headline $h --steps 4 --warmup 1 --no-demo-index --no-cli --no-cpu-baseline --no-end-to-end --no-proxy > gpurun_out/qp_$h.json 2> gpurun_out/qp_$h.err
	python -c "
import json; d=json.load(open('gpurun_out/qp_$h.json')); k=d['kernel_ms_per_step']; print('$h: %.0f reads/s  k_classify %.1f ms  seed %.1f  tail %.1f' % (d['value'], k['k_classify'], k.get('k_seed_scan', 0), k['wait_for_early_and_heavy_launches']))"
done
