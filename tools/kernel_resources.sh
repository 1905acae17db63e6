#!/bin/bash
# Register / scratch / LDS use of every kernel (compiler remarks; device-only compile, nothing is written into the tree).
# dsb_gpu.hip is compiled as its five units side by side (DSB_KUNIT), as the library is.   usage: tools/kernel_resources.sh [csrc-dir] [hipcc flags...]
src=${1:-$(dirname "$0")/../desamba_amd/csrc}; shift
inc=$(dirname "$0")/../include
out=$(mktemp -d)
for k in 0 1 2 3 4; do
	/opt/rocm/bin/hipcc -O3 -fno-strict-aliasing --offload-arch=gfx950 -std=c++17 -Wno-unused-value -I"$inc" -c --cuda-device-only -DDSB_KUNIT=$k "$@" \
		-Rpass-analysis=kernel-resource-usage "$src/dsb_gpu.hip" -o "$out/x$k.o" > "$out/r$k.txt" 2>&1 &
done
wait
cat "$out"/r?.txt |
	grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize|Occupancy|LDS Size" |
	sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - - | sed -E 's/Function Name: _Z[0-9]+([a-z_0-9]+[a-z])[0-9A-Z].*\tVGPRs:/\1\tVGPRs:/'
rm -rf "$out"
