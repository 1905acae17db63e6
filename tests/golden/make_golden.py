#!/usr/bin/env python3
"""Regenerates tests/golden/synth/* from the REFERENCE itself (needs /root/reference, i.e. this
container): small deterministic read sets (tools/readsim) and the SAM that the reference's
UB-pinned build (oracle/_ref/deSAMBA_ubfree, see oracle/Makefile) writes for them, plus the stock
reference's SAM for comparison.  Fixtures are data (inputs + expected outputs), no source."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "synth")
REF = os.path.join(ROOT, "oracle", "_ref")
DEMO = os.path.join(ROOT, "data", "demo")


def run(cmd):
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def strain():
    """tests/golden/strain: the second index (tools/make_strain_index.sh) and the reference's SAM for its 96 reads"""
    d = os.path.join(ROOT, "data", "strain"); out = os.path.join(ROOT, "tests", "golden", "strain")
    os.makedirs(out, exist_ok=True)
    subprocess.check_call([os.path.join(ROOT, "tools", "make_strain_index.sh"), d])
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", os.path.join(d, "index"), os.path.join(d, "reads.fq"), "-o", os.path.join(out, "reads.ubfree.sam")])
    open(os.path.join(out, "reads.fq.md5"), "w").write(hashlib.md5(open(os.path.join(d, "reads.fq"), "rb").read()).hexdigest() + "\n")


def kseq():
    """tests/golden/kseq: awkward FASTQ/FASTA texts of reads shorter than 40 bases (unmapped whatever the index, so the
    SAM_FULL output is the parser's view of name / sequence / quality) and the reference's output for each: pins the
    record rules of kseq_read (src/lib/utils.c:939-977) -- '\\r' kept, empty lines inside a sequence, whole-line
    quality, records with a quality string of the wrong length dropped -- and documents what the reference does
    with FASTA input (every other record lost on the first use of a slot)."""
    import random
    out = os.path.join(ROOT, "tests", "golden", "kseq"); os.makedirs(out, exist_ok=True)
    rnd = random.Random(5)
    dna = lambda k: bytes(rnd.choice(b"ACGTN") for _ in range(k))
    q = lambda k: bytes(rnd.choice(b"@>+5I#!") for _ in range(k))
    four = b"".join(b"@r%d some text\n%s\n+\n%s\n" % (i, s, q(len(s))) for i, s in enumerate(dna(rnd.randint(1, 38)) for _ in range(24)))
    multi = b""
    for i in range(16):
        lines = [dna(rnd.randint(1, 12)) for _ in range(rnd.randint(1, 3))]; tot = sum(map(len, lines)); qq = q(tot)
        cut = sorted(rnd.sample(range(1, tot), min(2, tot - 1))) if tot > 2 else []
        qs = [qq[a:b] for a, b in zip([0] + cut, cut + [tot])]
        multi += b"@m%d\n" % i + b"\n".join(lines) + b"\n+m%d\n" % i + b"\n".join(qs) + b"\n"
    blank = b"junk before\n\n" + four[:400].rsplit(b"\n@", 1)[0] + b"\n\n\n" + b"@b1\nACGT\n\nTTGA\n+\n55555555\n@b2\nAC\n+\n55\n\n\n"
    badqual = b"@g1\nACGTACGT\n+\n55555555\n@short\nACGTACGT\n+\n555\n@g2\nTTTT\n+\n5555\n@long\nACGT\n+\n555555\n@g3\nGGGG\n+\n5555\n"
    fasta = b"".join(b">c%d\n%s\n" % (i, dna(rnd.randint(5, 38))) for i in range(8))
    sets = {"four.fq": four, "crlf.fq": four.replace(b"\n", b"\r\n"), "multi.fq": multi, "blank.fq": blank, "badqual.fq": badqual,
            "no_nl.fq": four.rstrip(b"\n"), "records.fa": fasta}
    idx = os.path.join(DEMO, "index")
    for name, data in sets.items():
        path = os.path.join(out, name)
        open(path, "wb").write(data)
        run([os.path.join(REF, "deSAMBA"), "classify", "-t", "1", "-f", "SAM_FULL", idx, path, "-o", path + ".full.ref.sam"])


def ultralong():
    """tests/golden/synth/ultralong.fq.gz: ONE read of 0.88 Mbp -- longer than the 786432 bases beyond which the stock
    reference writes behind its per-thread 9-mer table (2^20 nodes, src/cly_mt.c:540-541; build_hash_table_M2,
    src/cly.c:2173-2224 needs 2^18 + L) -- made of the four longest demo genomes back to back, each with 12 % errors
    (numpy PCG64, seed 2026), and the SAM of the UB-pinned build, whose table holds 2^24 nodes (oracle/Makefile U7)."""
    import gzip
    import struct
    import numpy as np
    idx = os.path.join(DEMO, "index")
    with open(os.path.join(idx, "deSAMBA.ref_i"), "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        refs = [struct.unpack("<QQ", f.read(144)[128:]) for _ in range(n)]
    with open(os.path.join(idx, "deSAMBA.ref_b"), "rb") as f:
        nb = struct.unpack("<Q", f.read(8))[0]; txt = np.frombuffer(f.read(nb), dtype=np.uint8)
    rng = np.random.default_rng(2026)
    parts = []
    for (ln, off), take in zip(sorted(refs, reverse=True)[:4], (None, None, None, 100000)):
        pos = np.arange(off, off + (take or ln), dtype=np.int64)
        b = (txt[pos >> 2] >> (6 - 2 * (pos & 3))) & 3
        u = rng.random(len(b))
        sub = u < 0.05; b = np.where(sub, (b + rng.integers(1, 4, len(b))) & 3, b)
        keep = ~((u >= 0.05) & (u < 0.09))                               # deletions
        ins = (u >= 0.09) & (u < 0.12)                                   # an inserted base behind the base
        out = np.empty(2 * len(b), dtype=np.uint8); out[0::2] = b; out[1::2] = rng.integers(0, 4, len(b))
        mask = np.empty(2 * len(b), dtype=bool); mask[0::2] = keep; mask[1::2] = ins
        parts.append(out[mask])
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)].tobytes()
    fq = os.path.join(OUT, "ultralong.fq")
    with open(fq, "wb") as f:
        f.write(b"@ultralong_%d\n" % len(seq) + seq + b"\n+\n" + b"5" * len(seq) + b"\n")
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, fq, "-o", os.path.join(OUT, "ultralong.ubfree.sam")])
    with open(fq, "rb") as f, gzip.GzipFile(fq + ".gz", "wb", 9, mtime=0) as g:
        g.write(f.read())
    os.remove(fq)


def main():
    subprocess.check_call([os.path.join(ROOT, "tools", "make_demo_index.sh"), DEMO])
    if len(sys.argv) > 1 and sys.argv[1] == "ultralong":
        return ultralong()
    strain()
    kseq()
    ultralong()
    idx = os.path.join(DEMO, "index")
    sim = os.path.join(ROOT, "tools", "readsim")
    sets = [("ont20k", 12, 20000, 0.15, 11, "ont"), ("ngs150", 400, 150, 0.01, 12, "ngs"), ("pb", 24, 0, 0.13, 13, "pacbio"),
            ("ont5k_e25", 30, 5000, 0.25, 14, "ont")]
    for name, n, L, e, seed, prof in sets:
        fq = os.path.join(OUT, name + ".fq")
        run([sim, idx, fq, str(n), str(L), str(e), str(seed), prof])
        run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, fq, "-o", os.path.join(OUT, name + ".ubfree.sam")])
        run([os.path.join(REF, "deSAMBA"), "classify", "-t", "1", idx, fq, "-o", os.path.join(OUT, name + ".stock.sam")])
    # non-default output format and options (src/cly_mt.c:448-562)
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", "-f", "SAM_FULL", idx, os.path.join(OUT, "ngs150.fq"), "-o", os.path.join(OUT, "ngs150.full.ubfree.sam")])
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", "-l", "100", "-s", "30", "-r", "2", idx, os.path.join(OUT, "pb.fq"), "-o", os.path.join(OUT, "pb.l100s30r2.ubfree.sam")])
    for fmt, extra, name, outn in (("DES", [], "pb", "pb.des.ubfree.txt"), ("DES_FULL", ["-r", "1"], "ngs150", "ngs150.desfull.ubfree.txt"), ("DES", ["-r", "1"], "ont5k_e25", "ont5k_e25.des_r1.ubfree.txt")):
        run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", "-f", fmt] + extra + [idx, os.path.join(OUT, name + ".fq"), "-o", os.path.join(OUT, outn)])
    # the two heaviest reads of the 2000-read ONT set (tools/readsim seed 1, reads 1476 and 9): a tandem-repeat
    # region where one reference 9-mer matches dozens of read positions -> thousands of sparse-DP nodes
    heavy = os.path.join(OUT, "heavy.fq")
    if os.path.exists(heavy):
        run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, heavy, "-o", os.path.join(OUT, "heavy.ubfree.sam")])
        run([os.path.join(REF, "deSAMBA"), "classify", "-t", "1", idx, heavy, "-o", os.path.join(OUT, "heavy.stock.sam")])
    # read 2464 of the same generator (seed 1): its best chain starts at q = -1 (wrapped, unsigned), which the left
    # extension's predecessor tests compare as an unsigned number; and an NGS read hanging over the start of the
    # first reference, whose left extension works on negative (wrapped) reference coordinates
    wrapq = os.path.join(OUT, "wrapq.fq")
    if os.path.exists(wrapq):
        run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, wrapq, "-o", os.path.join(OUT, "wrapq.ubfree.sam")])
        run([os.path.join(REF, "deSAMBA"), "classify", "-t", "1", idx, wrapq, "-o", os.path.join(OUT, "wrapq.stock.sam")])
    # read 21607 of tools/readgen.c's first benchmark batch: 1202 anchors (more than the chain DP's LDS arrays hold), the slowest
    # read of that batch
    many = os.path.join(OUT, "manyanchors.fq")
    if os.path.exists(many):
        run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, many, "-o", os.path.join(OUT, "manyanchors.ubfree.sam")])
        run([os.path.join(REF, "deSAMBA"), "classify", "-t", "1", idx, many, "-o", os.path.join(OUT, "manyanchors.stock.sam")])
    # read 4346 of tools/readgen.c's first benchmark batch before that generator stopped running reads over the end of a
    # reference: the read wraps from the end of NC_003513.1 to its start, a hit hangs over the start of the reference, the
    # reference's unsigned window arithmetic (src/cly.c:2727) wraps and the STOCK binary dies with a segmentation fault;
    # the UB-pinned build reads such a window as zeros (oracle.h U6).  No stock golden for this one.
    overhang = os.path.join(OUT, "overhang.fq")
    if os.path.exists(overhang):
        run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, overhang, "-o", os.path.join(OUT, "overhang.ubfree.sam")])
    # the smallest reproducer of the stock reference's history dependence
    appc = os.path.join(OUT, "appc.fq")
    with open(appc, "w") as f:
        f.write("@r57271\nTTTATGTTTTATTTTTGGAATTGAAACTGATAGTGATTCTTTTGATGAATCATCATGTTTATGTTCATCAGATTTAGATCTGATTGAGTGTGAGAAGATTACTGTAAATGAAGCACCTAAATGTTTTGCAGAGTTTGAAAGACAGTGGGA\n+\n" + "5" * 150 + "\n")
        f.write("@r57290\nAATGCTCAGGTGGAGGAGGTCAGAGTGTATGATGGTACGGAGGAACTACCAGGGGATCCAGATATGATGAGATACATTGATAGATATGGTCAACACCAAACAAAAGATGCTGTAGAACAGGTGCTGCTTTATTAAGATGCTGTAGAACAGG\n+\n" + "5" * 151 + "\n")
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, appc, "-o", os.path.join(OUT, "appc.ubfree.sam")])
    run([os.path.join(REF, "deSAMBA"), "classify", "-t", "1", idx, appc, "-o", os.path.join(OUT, "appc.stock.sam")])
    # several input files in one run: max_read_l (src/cly.c:2958) runs over all of them (the per-thread buffers are allocated
    # once, before the loop over the files, src/cly_mt.c:538-556), so the 150-bp reads behind the 20-kbp file are filtered in
    # 3G mode -- one reference run over the five files, not the concatenation of five runs
    # (ngs_e14: 150-bp reads at 14 % error, some of which score between the 2G and the short-3G cut)
    run([sim, idx, os.path.join(OUT, "ngs_e14.fq"), "120", "150", "0.14", "21", "ngs"])
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx, os.path.join(OUT, "ngs_e14.fq"), "-o", os.path.join(OUT, "ngs_e14.ubfree.sam")])
    multi = [os.path.join(OUT, n + ".fq") for n in ("ont20k", "ngs_e14", "pb", "appc", "wrapq", "ngs150")]
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx] + multi + ["-o", os.path.join(OUT, "multi6.ubfree.sam")])
    multi = [os.path.join(OUT, n + ".fq") for n in ("pb", "ngs_e14", "ngs150", "appc")]
    run([os.path.join(REF, "deSAMBA_ubfree"), "classify", "-t", "1", idx] + multi + ["-o", os.path.join(OUT, "multi4.ubfree.sam")])
    # demo: the reference's own quick-start output (README.md:30-42): md5 + first 60 lines
    demo_sam = os.path.join(DEMO, "ref_demo.sam")
    run([os.path.join(REF, "deSAMBA"), "classify", "-t", "4", idx, os.path.join(DEMO, "ERR1050068.fastq"), "-o", demo_sam])
    data = open(demo_sam, "rb").read()
    with open(os.path.join(ROOT, "tests", "golden", "demo_sam.md5"), "w") as f:
        f.write(hashlib.md5(data).hexdigest() + "\n")
    with open(os.path.join(ROOT, "tests", "golden", "demo_head60.sam"), "wb") as f:
        f.write(b"".join(data.splitlines(True)[:60]))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    sys.exit(main())
