#!/bin/bash
# Second test index (tests/golden/strain): 4 Mbp of synthetic strains with mobile elements and tandem repeats
# (tools/synth_ref.py, seed 7, up to 8 repeats of unit length 2-4 per genome), indexed as in make_demo_index.sh (this repo's builder on a GPU box, else the reference binary), plus 96 ONT reads of 15 kbp
# simulated from it (tools/readsim, seed 21).  Idempotent; ~25 s, ~2 GB RSS.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/data/strain}"
if [ -f "$OUT/index/deSAMBA.ref_p" ] && [ -f "$OUT/reads.fq" ]; then exit 0; fi
mkdir -p "$OUT/index"
[ -x "$ROOT/oracle/_ref/kmer_srt" ] || make -C "$ROOT/oracle" tools
[ -x "$ROOT/oracle/_ref/deSAMBA" ] || { echo "oracle/_ref/deSAMBA missing (run make -C oracle ref where /root/reference exists)"; exit 1; }
[ -x "$ROOT/tools/readsim" ] || gcc -O2 -o "$ROOT/tools/readsim" "$ROOT/tools/readsim.c" -lm
python3 "$ROOT/tools/synth_ref.py" "$OUT/syn.fa" 4 7 9 5 2>/dev/null
if ! { [ -x "$ROOT/desamba_amd/bin/deSAMBA" ] && "$ROOT/desamba_amd/bin/deSAMBA" index "$OUT/syn.fa" "$OUT/index" >/dev/null 2>&1; }; then
	# no GPU here: the reference binary (same bytes, tests/test_index_build.py)
	rm -rf "$OUT/index"; mkdir -p "$OUT/index"
	"$ROOT/oracle/_ref/kmer_srt" "$OUT/syn.fa" "$OUT/kmer.srt" 2>/dev/null
	"$ROOT/oracle/_ref/deSAMBA" index "$OUT/kmer.srt" "$OUT/syn.fa" "$OUT/index" >/dev/null 2>&1
	rm -f "$OUT/kmer.srt"
fi
"$ROOT/tools/readsim" "$OUT/index" "$OUT/reads.fq" 96 15000 0.15 21 ont >/dev/null 2>&1
ls "$OUT/index" | wc -l
