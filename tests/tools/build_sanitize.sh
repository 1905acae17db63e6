#!/bin/bash
# The index builder's stages (dsb_build_impl.h in one piece, dsb_build_parts.h in ranges of prefixes) as the host emulation under
# AddressSanitizer + UBSan over three golden references; digests compared with the reference's.   tests/tools/build_sanitize.sh
set -e -o pipefail
cd "$(dirname "$0")/../.."
tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-strict-aliasing -fPIC -shared -Idesamba_amd/csrc -Iinclude -o "$tmp/libdsbemu_build_san.so" tests/emu/emu_build.cpp -lz
cat > "$tmp/run.py" <<PY
import ctypes as C, sys, tempfile, json
sys.path.insert(0, "tests")
import build_lib
L = C.CDLL("$tmp/libdsbemu_build_san.so")
L.dsb_emu_index_build_parts.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
L.dsb_emu_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]
G = "tests/golden/build"; bad = 0
for name, parts in (("graph3", 7), ("graph1", 3), ("reader", 2)):
    want = json.load(open(G + "/" + name + ".md5.json"))
    d = tempfile.mkdtemp(dir="$tmp"); st = (C.c_uint64 * 11)()
    rc = L.dsb_emu_index_build_parts(None, (G + "/" + name + ".fa.gz").encode(), d.encode(), 1 << 40, parts, st)
    ok = rc == 0 and build_lib.digest_dir(d) == want; bad += not ok; print(name, "in", parts, "ranges:", "ok" if ok else "WRONG")
    d = tempfile.mkdtemp(dir="$tmp"); st = (C.c_uint64 * 4)()
    rc = L.dsb_emu_index_build(None, (G + "/" + name + ".fa.gz").encode(), d.encode(), st)
    ok = rc == 0 and build_lib.digest_dir(d) == want; bad += not ok; print(name, "in one piece:", "ok" if ok else "WRONG")
sys.exit(1 if bad else 0)
PY
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python3 "$tmp/run.py" 2>&1 | tee "$tmp/log"
if grep -q "runtime error\|AddressSanitizer\|WRONG" "$tmp/log"; then echo "findings above"; exit 1; fi
echo "clean"
