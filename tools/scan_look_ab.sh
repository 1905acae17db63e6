#!/bin/bash
# k_seed_scan's look-ahead (DSB_SCAN_LOOK=after_seed,back_fwd,fwd_n,stride_n) on the headline / demo index: time and probes issued per setting
#   tools/scan_look_ab.sh lib.so headline "<settings...>"
cd "$(dirname "$0")/.."
L=$1; H=$2; shift 2
for v in $*; do
	if [ $v = default ]; then unset DSB_SCAN_LOOK; else export DSB_SCAN_LOOK=$v; fi
	DSB_LIB_PATH=$PWD/$L python bench.py --headline $H --steps 3 --warmup 1 --batches 1 --no-demo-index --no-cli --no-cpu-baseline --no-end-to-end --no-short-reads --no-proxy --no-seed-hbm --no-budget-build > gpurun_out/sl.json 2> gpurun_out/sl.err
	python -c "
import json; d=json.load(open('gpurun_out/sl.json')); r=d['roofline_seed_lookup']; print('$H $v: k_seed_scan %.1f ms  issued %.1f GB  k_classify %.1f ms  value %.0f' % (r['ms'], r['algorithmic_bytes'] / 1e9, d['kernel_ms_per_step']['k_classify'], d['value']))"
done
