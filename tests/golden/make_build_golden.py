#!/usr/bin/env python3
"""Regenerates tests/golden/build/*: small reference FASTA files (synthetic, seeded) and, for each, the md5 of every
index file the REFERENCE binary writes for it (oracle/_ref/kmer_srt -> oracle/_ref/deSAMBA index; needs
/root/reference to have been compiled by oracle/Makefile, i.e. this container).  Fixtures are data: FASTA inputs and
expected digests.  `.bwt` is digested with the bytes behind its last symbol zeroed when the index has fewer than 257
blocks (the reference leaves uninitialised heap there, src/bwt.c:222-238); `.ref_i` field-wise (name, length, offset)."""
import gzip
import hashlib
import json
import os
import random
import struct
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import build_lib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "build")
REF = os.path.join(ROOT, "oracle", "_ref")


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def graph_case(seed, n_ref, base_len):
    """references cut from three random sources with point mutations (branching), reverse complements, N runs, tandem
    repeats, homopolymer tails, lower case, ACGT runs of exactly 31 and 30 bases, a 10-base record, an empty record,
    an exact duplicate"""
    r = random.Random(seed)
    pool = ["".join(r.choice("ACGT") for _ in range(base_len)) for _ in range(3)]
    recs = []
    for i in range(n_ref):
        src = pool[i % 3]
        a = r.randrange(0, base_len // 2); b = a + r.randrange(200, base_len // 2)
        s = list(src[a:b])
        for _ in range(len(s) // 200):
            s[r.randrange(len(s))] = r.choice("ACGT")
        s = "".join(s)
        if i % 4 == 1: s = rc(s)
        if i % 5 == 2: s = s[:300] + "N" * r.randrange(1, 40) + s[300:]
        if i % 5 == 3: s = s[:500] + "ACGT" * 30 + s[500:] + "A" * 50
        if i % 6 == 4: s = s.lower()
        if i % 7 == 5: s = s[:100] + "NNN" + s[100:131] + "NN" + s[131:161] + "RYK" + s[161:]
        recs.append(("ref%d some comment" % i, s))
    recs += [("tiny", "ACGTACGTAC"), ("empty", ""), ("dup", recs[0][1][:400])]
    out = []
    for n, s in recs:
        out.append(">%s\n" % n)
        w = r.choice([60, 70, 80])
        out += [s[k:k + w] + "\n" for k in range(0, len(s), w)]
    return "".join(out).encode()


def reader_case():
    """what the reference's FASTA reader does with awkward text (src/lib/utils.c:939-977): '\\r\\n' line ends, an empty
    line inside a sequence (puts a newline into the text and swallows the next line, header or not), '@' headers, a
    FASTQ record, tabs in header lines, text before the first header, no newline at the end"""
    r = random.Random(77)
    dna = lambda k: "".join(r.choice("ACGT") for _ in range(k))
    t = "junk before the first header\n"
    t += ">crlf with comment\r\n" + dna(70) + "\r\n" + dna(64) + "\r\n"
    t += ">blank\tline inside\n" + dna(80) + "\n\n>swallowed header\n" + dna(75) + "\n"
    t += "@atsign\n" + dna(90) + "\n" + dna(50) + "\n"
    s = dna(120)
    t += "@fastq record\n" + s + "\n+\n" + "I" * 120 + "\n"
    t += ">after fastq\n" + dna(100) + "\n" + dna(33) + "\n"
    # (enough ordinary sequence behind it for the reference's builder to work at all: its '$' goes to the first unitig
    # start among the first sixteenth of the sorted k-mers, src/idx.c:733,781-793 -- a handful of k-mers may have none)
    return t.encode() + graph_case(5, 8, 4000).rstrip(b"\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {"graph1": graph_case(1, 12, 9000), "graph2": graph_case(2, 12, 9000), "graph3": graph_case(3, 40, 60000), "reader": reader_case()}
    for name, text in cases.items():
        fa = os.path.join(OUT, name + ".fa.gz")
        with open(fa, "wb") as f:
            f.write(gzip.compress(text, mtime=0))
        with tempfile.TemporaryDirectory() as tmp:
            plain = os.path.join(tmp, "ref.fa"); open(plain, "wb").write(text)
            srt = os.path.join(tmp, "kmer.srt")
            # (the k-mer list of the reader case comes from this repo's builder: oracle/_ref/kmer_srt reads FASTA the ordinary way)
            if name == "reader":
                build_lib.write_kmer_srt_from_text(build_lib.reader_view(text), srt)
            else:
                subprocess.run([os.path.join(REF, "kmer_srt"), plain, srt], check=True, stderr=subprocess.DEVNULL)
            subprocess.run([os.path.join(REF, "deSAMBA"), "index", srt, plain, os.path.join(tmp, "idx")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            json.dump(build_lib.digest_dir(os.path.join(tmp, "idx")), open(os.path.join(OUT, name + ".md5.json"), "w"), indent=1, sort_keys=True)
        print(name, os.path.getsize(fa))


if __name__ == "__main__":
    main()
