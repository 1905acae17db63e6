#!/usr/bin/env python3
"""Regenerates tests/golden/eklevel/*: exist-k-mer filter tables of every k the reference knows (src/idx.c:966-982) on one small
reference.  The reference picks the table size -- and with it k = 16 .. 20 and the 30- .. 37-bit hash mask -- from the number of
31-mers (get_EXIST_kmer, src/idx.c:986-996); only k = 16 and 18 ever came up on the indexes of rounds 1-3.  Here the UB-pinned
build of the reference (oracle/_ref/deSAMBA_ubfree, whose `index` takes DSB_FORCE_EK_LEVEL: oracle/Makefile) builds the index of
a 300-kbp reference at level 1 (k = 17, 2 x 256 MiB), 5 (k = 19, 2 x 4 GiB) and 7 (k = 20, 2 x 16 GiB) and classifies 150 reads on
each.  Fixtures are data: the reference FASTA, the reads, per level the md5 of every index file and the SAM.  Level 7 needs
~34 GiB of memory and of disk for a few minutes.   usage: make_eklevel_golden.py [levels...]"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "eklevel")
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import build_lib


def md5_file(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    levels = [int(x) for x in sys.argv[1:]] or [1, 5, 7]
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="eklevel_", dir=os.environ.get("TMPDIR", "/tmp"))
    fa = os.path.join(tmp, "ref.fa"); fq = os.path.join(tmp, "reads.fq")
    if not os.path.exists(os.path.join(OUT, "ref.fa.gz")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, "0.3", "5"], stderr=subprocess.DEVNULL)
        with open(fa, "rb") as f, gzip.GzipFile(os.path.join(OUT, "ref.fa.gz"), "wb", mtime=0) as g:
            g.write(f.read())
    with gzip.open(os.path.join(OUT, "ref.fa.gz"), "rb") as g, open(fa, "wb") as f:
        f.write(g.read())
    subprocess.check_call([os.path.join(REF, "kmer_srt"), fa, os.path.join(tmp, "kmer.srt")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    digests = {}
    if os.path.exists(os.path.join(OUT, "digests.json")):
        digests = json.load(open(os.path.join(OUT, "digests.json")))
    for lv in levels:
        d = os.path.join(tmp, "idx%d" % lv); os.makedirs(d)
        env = dict(os.environ, DSB_FORCE_EK_LEVEL=str(lv))
        subprocess.check_call([os.path.join(REF, "deSAMBA_ubfree"), "index", os.path.join(tmp, "kmer.srt"), fa, d], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if not os.path.exists(os.path.join(OUT, "reads.fq.gz")):
            # 100 ONT-like reads of 3 kbp at 12 % error, 50 short reads at 2 % (fast and slow path, both strands)
            subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), d, fq, "100", "3000", "0.12", "7", "ont"], stdout=subprocess.DEVNULL)
            subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), d, fq + ".b", "50", "250", "0.02", "8", "ngs"], stdout=subprocess.DEVNULL)
            with gzip.GzipFile(os.path.join(OUT, "reads.fq.gz"), "wb", mtime=0) as g:
                g.write(open(fq, "rb").read() + open(fq + ".b", "rb").read())
        with gzip.open(os.path.join(OUT, "reads.fq.gz"), "rb") as g, open(fq, "wb") as f:
            f.write(g.read())
        sam = os.path.join(OUT, "level%d.ubfree.sam" % lv)
        subprocess.check_call([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "4", d, fq, "-o", sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dg = {e: hashlib.md5(build_lib.canonical_bytes(d, e)).hexdigest() for e in build_lib.EXTS if e not in (".exk0", ".exk1")}
        dg[".exk0"] = md5_file(os.path.join(d, "deSAMBA.exk0")); dg[".exk1"] = md5_file(os.path.join(d, "deSAMBA.exk1"))
        dg["exk_bytes"] = os.path.getsize(os.path.join(d, "deSAMBA.exk0"))
        digests[str(lv)] = dg
        json.dump(digests, open(os.path.join(OUT, "digests.json"), "w"), indent=1, sort_keys=True)
        shutil.rmtree(d)
        print("level", lv, "done:", dg["exk_bytes"], "bytes per table,", sum(1 for _ in open(sam)), "SAM lines", flush=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
