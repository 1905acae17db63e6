#!/bin/bash
# The device code of k_classify as the 1-lane host emulation (tests/emu), built with AddressSanitizer + UBSan, over the golden read
# sets, the edge cases and 150 fresh 50-kbp reads (GPU sanitizers are not available on the pool: this is the CPU build of the same
# header).  Leaves tests/emu/libdsbemu.so as it was.     tests/tools/emu_sanitize.sh
set -e -o pipefail
cd "$(dirname "$0")/../.."
tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-strict-aliasing -fPIC -shared -DDSB_HOST_EMU -Idesamba_amd/csrc -Iinclude -Itests/emu \
	-o "$tmp/libdsbemu.so" tests/emu/emu_classify.cpp desamba_amd/csrc/dsb_index.cpp
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
	DSB_EMU_LIB="$tmp/libdsbemu.so" python3 tests/tools/emu_sanitize.py 2>&1 | tee "$tmp/log"
if grep -q "runtime error\|AddressSanitizer" "$tmp/log"; then echo "sanitizer findings above"; exit 1; fi
echo "clean"
