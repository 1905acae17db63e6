#!/bin/bash
# The distance between the stock reference and the UB-pinned canonical semantics, rule by rule: N synthetic 50-kbp ONT
# reads (bench.py's batch 0: seed 1000) on an index, everything at -t 1 (the stock binary's result depends on its thread schedule):
#   stock -t 1            the reference as it is
#   ubfree                all pins (U1 read-buffer pads, U2 window pads, U5 lv_extd fronts, U6 windows beyond the text)
#   u1 / u2 / u5          one pin alone (+ U6, without which the stock code reads far behind the text on small indexes)
#   midcall               all pins, the middle-gap window filled once per sdp_middle_M2 call instead of once per gap
# Prints differing reads (any SAM digit) and flag/reference changes of every variant against stock -t 1, and stock -t 1 against stock -t 8.
#   tests/tools/residue.sh <IndexDir> [n_reads] [outdir]        (CPU only; needs oracle/_ref built with /root/reference present)
cd "$(dirname "$0")/../.."
I=$1; N=${2:-8192}; OUT=${3:-/tmp/residue}; mkdir -p "$OUT"
make -s -C oracle ref ref_ubfree > /dev/null
make -s -C oracle ref_variant NAME=u1 VSED='$(UB_U1) $(UB_U6)' > /dev/null; make -s -C oracle ref_variant NAME=u2 VSED='$(UB_U2) $(UB_U6)' > /dev/null
make -s -C oracle ref_variant NAME=u5 VSED='$(UB_U5) $(UB_U6)' > /dev/null; make -s -C oracle ref_variant NAME=midcall VSED='$(UB_U1) $(UB_U2_MIDCALL) $(UB_U5) $(UB_U6)' > /dev/null
python3 tools/gen_fastq.py "$I" "$OUT/r.fq" "$N" 50000 0.15 1000 ont 8 > /dev/null
T=$(tests/tools/host_cpus.sh)
# the variants are thread-invariant (checked below for two of them): they run on all cores; the stock binary at -t 1 and at -t T
for v in ubfree u1 u2 u5 midcall; do oracle/_ref/deSAMBA_$v classify -t $T "$I" "$OUT/r.fq" -o "$OUT/$v.sam" > /dev/null 2>&1; done
oracle/_ref/deSAMBA_ubfree classify -t 1 "$I" "$OUT/r.fq" -o "$OUT/ubfree_t1.sam" > /dev/null 2>&1 &
oracle/_ref/deSAMBA_midcall classify -t 1 "$I" "$OUT/r.fq" -o "$OUT/midcall_t1.sam" > /dev/null 2>&1 &
oracle/_ref/deSAMBA classify -t 1 "$I" "$OUT/r.fq" -o "$OUT/stock_t1.sam" > /dev/null 2>&1 &
wait
oracle/_ref/deSAMBA classify -t $T "$I" "$OUT/r.fq" -o "$OUT/stock_tN.sam" > /dev/null 2>&1
python3 - "$OUT" "$T" <<'PY'
import sys
out, T = sys.argv[1], sys.argv[2]
def by_read(p):
    d = {}
    for ln in open(p, "rb"):
        d.setdefault(ln.split(b"\t", 1)[0], []).append(ln)
    return d
def dist(a, b):
    diff = [k for k in a if a[k] != b.get(k)]
    fr = [k for k in a if [l.split(b"\t")[1:3] for l in a[k]] != [l.split(b"\t")[1:3] for l in b.get(k, [])]]
    return len(diff), len(fr), fr
stock = by_read(out + "/stock_t1.sam"); n = len(stock)
print("%d reads; against stock -t 1: differing reads (any digit) / reads whose (flag, reference) list differs" % n)
for v in ("stock_tN", "ubfree", "u1", "u2", "u5", "midcall"):
    d, f, fr = dist(stock, by_read("%s/%s.sam" % (out, v)))
    print("  %-9s %5d (%.2f %%)  %3d   %s" % (v if v != "stock_tN" else "stock -t " + T, d, 100.0 * d / n, f, b" ".join(fr[:6]).decode()))
for v in ("ubfree", "midcall"):
    d, f, _ = dist(by_read("%s/%s.sam" % (out, v)), by_read("%s/%s_t1.sam" % (out, v)))
    print("  thread invariance of %s (-t %s vs -t 1): %d differing reads" % (v, T, d))
PY
