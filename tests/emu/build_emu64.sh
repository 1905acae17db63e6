#!/bin/bash
# tests/emu/libdsbemu64.so: the device code as 64 lanes with a race detector (emu_simt.cpp).  The device code is compiled with
# -fsanitize=thread for its instrumentation hooks only: libtsan is NOT linked, emu_simt.cpp implements the hooks.
set -e
cd "$(dirname "$0")/../.."
tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
I="-Idesamba_amd/csrc -Iinclude -Itests/emu"
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-strict-aliasing -fPIC -DDSB_HOST_EMU -DDSB_EMU_LANES=64 $I -c tests/emu/emu_classify.cpp -o "$tmp/classify.o" &
g++ -std=c++17 -O2 -g -fPIC -c tests/emu/emu_simt.cpp -o "$tmp/simt.o" &
g++ -std=c++17 -O2 -fno-strict-aliasing -fPIC -DDSB_HOST_EMU $I -c desamba_amd/csrc/dsb_index.cpp -o "$tmp/index.o" &
wait
g++ -shared -o tests/emu/libdsbemu64.so "$tmp/classify.o" "$tmp/simt.o" "$tmp/index.o" -ldl
