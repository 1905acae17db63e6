#!/bin/bash
# Build the demo index under data/demo from the committed fixtures
# tests/golden/demo/*.zip.  Index CONSTRUCTION is outside this repo's scope: the
# reference binary built by oracle/Makefile does it, fed by oracle/_ref/kmer_srt instead of Jellyfish.
# Idempotent; ~25 s, ~2 GB RSS.
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
[ -x "$ROOT/oracle/_ref/kmer_srt" ] || make -C "$ROOT/oracle" tools
[ -x "$ROOT/oracle/_ref/deSAMBA" ] || { echo "oracle/_ref/deSAMBA missing (run make -C oracle ref where /root/reference exists)"; exit 1; }
"$ROOT/oracle/_ref/kmer_srt" "$OUT/viral-gs.fa" "$OUT/kmer.srt" 2>/dev/null
"$ROOT/oracle/_ref/deSAMBA" index "$OUT/kmer.srt" "$OUT/viral-gs.fa" "$OUT/index" >/dev/null 2>&1
rm -f "$OUT/kmer.srt"
ls "$OUT/index" | wc -l
