#!/bin/bash
# Index construction at scale on a GPU box: a synthetic reference collection of <Mbp> million bases (tools/synth_ref.py)
# is indexed by the reference binary (k-mer list by oracle/_ref/kmer_srt) and by `deSAMBA index` of this repo; every
# file is compared byte for byte (.ref_i: names, lengths, offsets).
#   tests/tools/big_build.sh [outdir] [Mbp] [synth_ref args...]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; MBP=${2:-380}; shift 2
D=data/bigb; mkdir -p "$OUT" $D
TIMEFORMAT="%R"
python3 tools/synth_ref.py $D/syn.fa $MBP ${@:-1} 2>&1
tg=$( { time desamba_amd/bin/deSAMBA index $D/syn.fa $D/own > /dev/null 2> "$OUT/bigb_own.log"; } 2>&1 )
echo "this repo: ${tg}s wall;  $(tail -2 "$OUT/bigb_own.log" | tr '\n' ' ')"
t1=$( { time oracle/_ref/kmer_srt $D/syn.fa $D/kmer.srt > "$OUT/bigb_kmer.log" 2>&1; } 2>&1 )
echo "k-mer list (oracle/_ref/kmer_srt, one thread): ${t1}s"
t2=$( { time oracle/_ref/deSAMBA index $D/kmer.srt $D/syn.fa $D/ref > "$OUT/bigb_ref.log" 2>&1; } 2>&1 )
echo "reference deSAMBA index: ${t2}s"
python3 - $D/own $D/ref <<'PY'
import sys
sys.path.insert(0, "tests")
import build_lib
a, b = sys.argv[1], sys.argv[2]
bad = 0
for e in build_lib.EXTS:
    x, y = build_lib.canonical_bytes(a, e), build_lib.canonical_bytes(b, e)
    same = x == y
    bad += not same
    print("%-7s %12d bytes  %s" % (e, len(y), "identical" if same else "DIFFERENT"))
print("index files identical" if not bad else "%d FILES DIFFER" % bad)
PY
rm -rf $D
