#!/bin/bash
# Build the demo index under data/demo from the committed fixtures
# tests/golden/demo/*.zip: with `deSAMBA index` of this repo where there is a GPU (< 1 s), else with the reference
# binary built by oracle/Makefile, fed by oracle/_ref/kmer_srt instead of Jellyfish (~25 s, ~2 GB RSS).  The two write
# the same bytes (tests/test_index_build.py).  Idempotent.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/data/demo}"
if [ -f "$OUT/index/deSAMBA.ref_p" ] && [ -f "$OUT/ERR1050068.fastq" ]; then exit 0; fi
mkdir -p "$OUT/index"
python3 - "$ROOT" "$OUT" <<'PY'
import sys, zipfile
root, out = sys.argv[1], sys.argv[2]
for z in ("viral-gs.zip", "ERR1050068.zip"):
    zipfile.ZipFile(root + "/tests/golden/demo/" + z).extractall(out)
PY
if [ -x "$ROOT/desamba_amd/bin/deSAMBA" ] && "$ROOT/desamba_amd/bin/deSAMBA" index "$OUT/viral-gs.fa" "$OUT/index" >/dev/null 2>&1; then
	ls "$OUT/index" | wc -l
	exit 0
fi
rm -rf "$OUT/index"; mkdir -p "$OUT/index"
[ -x "$ROOT/oracle/_ref/kmer_srt" ] || make -C "$ROOT/oracle" tools
[ -x "$ROOT/oracle/_ref/deSAMBA" ] || { echo "oracle/_ref/deSAMBA missing (run make -C oracle ref where /root/reference exists)"; exit 1; }
"$ROOT/oracle/_ref/kmer_srt" "$OUT/viral-gs.fa" "$OUT/kmer.srt" 2>/dev/null
"$ROOT/oracle/_ref/deSAMBA" index "$OUT/kmer.srt" "$OUT/viral-gs.fa" "$OUT/index" >/dev/null 2>&1
rm -f "$OUT/kmer.srt"
ls "$OUT/index" | wc -l
