#!/usr/bin/env python3
"""Regenerates tests/golden/analysis/*: a synthetic taxonomy (nodes.dmp layout: `tid | parent | rank | ...`) over the taxids
of the demo reference, and what the REFERENCE binary prints for `analysis ana_meta` / `analysis ana_meta_base` on SAM
files of the golden sets (oracle/_ref/deSAMBA, i.e. this container).  Fixtures are data: the taxonomy, and expected
outputs; the SAM inputs are tests/golden/synth/*.sam, tests/golden/demo_head60.sam and the hand-made
tests/golden/analysis/tricky.sam (same-score strain / species pairs, a first record without score, an unknown taxid, '@'
lines, minimap2 tags, a multi-record read at the end of the file)."""
import os
import re
import subprocess
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "analysis")
REF = os.path.join(ROOT, "oracle", "_ref", "deSAMBA")
SAMS = ["analysis/tricky.sam", "demo_head60.sam", "synth/pb.ubfree.sam", "synth/ngs150.ubfree.sam", "synth/ont20k.ubfree.sam", "synth/appc.ubfree.sam", "synth/multi6.ubfree.sam"]


def taxonomy():
    """species under 41 genera under 7 families under 3 orders under 'Viruses' (10239) under the root; every fifth taxid
    of the reference is a 'no rank' strain whose parent is the previous taxid of the list (a hit on the strain and one on
    its species with the same score resolve to the strain, ana_get_tid src/analysis.c:1271-1330)"""
    fa = zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", "demo", "viral-gs.zip")).read("viral-gs.fa")
    tids = sorted({int(m) for m in re.findall(rb"^>tid\|(\d+)\|", fa, re.M)})
    rows = {1: (1, "no rank"), 10239: (1, "superkingdom")}
    for k in range(3):
        rows[3200000 + k] = (10239, "order")
    for k in range(7):
        rows[3100000 + k] = (3200000 + k % 3, "family")
    for k in range(41):
        rows[3000000 + k] = (3100000 + k % 7, "genus")
    for i, t in enumerate(tids):
        if i % 5 == 4:
            rows[t] = (tids[i - 1], "no rank")
        else:
            rows[t] = (3000000 + t % 41, "species")
    with open(os.path.join(OUT, "nodes.dmp"), "w") as f:
        for t in sorted(rows):
            p, r = rows[t]
            f.write("%d\t|\t%d\t|\t%s\t|\t\t|\t0\t|\t1\t|\t11\t|\t1\t|\t0\t|\t1\t|\t0\t|\t0\t|\t\t|\n" % (t, p, r))


def main():
    os.makedirs(OUT, exist_ok=True)
    taxonomy()
    for sam in SAMS:
        src = os.path.join(ROOT, "tests", "golden", sam)
        for cmd in ("ana_meta", "ana_meta_base"):
            # the reference prints the path of its temporary file: run it on a fixed relative name
            tmp = os.path.join(OUT, "in.sam")
            open(tmp, "wb").write(open(src, "rb").read())
            p = subprocess.run([REF, "analysis", cmd, "in.sam", "nodes.dmp"], cwd=OUT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
            open(os.path.join(OUT, os.path.basename(sam).replace(".sam", "") + "." + cmd + ".txt"), "wb").write(p.stdout)
            os.remove(tmp)
            if os.path.exists(tmp + ".temp"):
                os.remove(tmp + ".temp")
    # the FASTQ helpers of the same usage text on the reader fixtures (tests/golden/kseq): stdout + the summary line on stderr
    kd = os.path.join(ROOT, "tests", "golden", "kseq")
    for fq in sorted(f for f in os.listdir(kd) if f.endswith((".fq", ".fa"))):
        for tag, args in (("count_base", ["count_base", fq]), ("fastq_to_fasta", ["fastq_to_fasta", fq]), ("split_1_2", ["split_fastq", fq, "1", "2"]), ("split_0_3", ["split_fastq", fq, "0", "3"])):
            p = subprocess.run([REF, "analysis"] + args, cwd=kd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            err = b"".join(l for l in p.stderr.splitlines(True) if b"read number:" in l)
            open(os.path.join(OUT, "%s.%s.txt" % (fq, tag)), "wb").write(p.stdout + b"--- stderr ---\n" + err)
    print(sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
