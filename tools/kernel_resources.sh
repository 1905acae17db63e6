#!/bin/bash
# Register / scratch / LDS use of every kernel (compiler remarks; device-only compile, nothing is written into the tree).
# usage: tools/kernel_resources.sh [csrc-dir]
src=${1:-$(dirname "$0")/../desamba_amd/csrc}
inc=$(dirname "$0")/../include
out=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -fno-strict-aliasing --offload-arch=gfx950 -std=c++17 -Wno-unused-value -I"$inc" -c --cuda-device-only \
	-Rpass-analysis=kernel-resource-usage "$src/dsb_gpu.hip" -o "$out/x.o" 2>&1 |
	grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize|Occupancy|LDS Size" |
	sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - - | sed -E 's/Function Name: _Z[0-9]+([a-z_0-9]+[a-z])[0-9A-Z].*\tVGPRs:/\1\tVGPRs:/'
rm -rf "$out"
