#!/usr/bin/env python3
"""CPU-only differential: the device code compiled for the host (tests/emu) against the oracle on the reads
[lo, hi) of a FASTQ, history = running max of read lengths as in a single-threaded run.  TEST TOOL.
    python tests/tools/emu_diff.py <reads.fq> [lo hi]      (index: $DSB_INDEX, default the demo index)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import emu_lib, oracle_lib
import desamba_amd as D
recs = D.read_fastq(sys.argv[1])
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0; hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(recs)
idx = os.environ.get("DSB_INDEX", os.path.join(ROOT, "data", "demo", "index"))
emu = emu_lib.Emu(idx); ora = oracle_lib.Oracle(idx)
hist = max([len(r[1]) for r in recs[:lo]] + [0]); bad = 0; t = time.time()
for i in range(lo, min(hi, len(recs))):
    nm, seq, q = recs[i]
    if emu.classify(seq, hist) != ora.classify(seq, hist):
        bad += 1; print("DIFF", i, nm, flush=True)
    hist = max(hist, len(seq))
print("checked %d reads [%d,%d): %d differ, %.0f s" % (min(hi, len(recs)) - lo, lo, hi, bad, time.time() - t))
