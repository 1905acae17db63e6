"""`deSAMBA analysis ana_meta | ana_meta_base`: the taxonomy roll-up of a classify result, compared byte for
byte with what the reference binary prints (tests/golden/analysis/*.txt, made by tests/golden/make_analysis_golden.py
on a synthetic taxonomy over the taxids of the demo reference).  Host-side text processing: no GPU needed."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

CLI = os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA")
ANA = os.path.join(GOLDEN, "analysis")
CASES = [("tricky", "analysis/tricky.sam"), ("demo_head60", "demo_head60.sam"), ("pb.ubfree", "synth/pb.ubfree.sam"), ("ngs150.ubfree", "synth/ngs150.ubfree.sam"),
         ("ont20k.ubfree", "synth/ont20k.ubfree.sam"), ("appc.ubfree", "synth/appc.ubfree.sam"), ("multi6.ubfree", "synth/multi6.ubfree.sam")]


def run(cmd, sam, tmp_path):
    # (the output names the reference's temporary file "<SAM>.temp": same relative name as when the golden was made)
    shutil.copy(os.path.join(GOLDEN, sam), tmp_path / "in.sam")
    shutil.copy(os.path.join(ANA, "nodes.dmp"), tmp_path / "nodes.dmp")
    return subprocess.run([CLI, "analysis", cmd, "in.sam", "nodes.dmp"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.parametrize("cmd", ["ana_meta", "ana_meta_base"])
@pytest.mark.parametrize("name,sam", CASES)
def test_analysis_output_is_the_reference_s(built, name, sam, cmd, tmp_path):
    """read counts (ana_meta) and MAPQ-weighted bases (ana_meta_base) per taxon, rolled up the tree: same-score strain /
    species pairs, records without score, unknown taxids, '@' lines, minimap2 tags, a read cut off by the end of the file
    (tricky.sam), and the SAM of six golden read sets"""
    p = run(cmd, sam, tmp_path)
    assert p.returncode == 0, p.stderr
    assert p.stdout == open(os.path.join(ANA, "%s.%s.txt" % (name, cmd)), "rb").read()


KSEQ = os.path.join(GOLDEN, "kseq")


@pytest.mark.parametrize("fq", sorted(f for f in os.listdir(KSEQ) if f.endswith((".fq", ".fa"))))
@pytest.mark.parametrize("tag,args", [("count_base", ["count_base"]), ("fastq_to_fasta", ["fastq_to_fasta"]), ("split_1_2", ["split_fastq", "1", "2"]), ("split_0_3", ["split_fastq", "0", "3"])])
def test_fastq_helpers_of_analysis(built, fq, tag, args, tmp_path):
    """count_base / fastq_to_fasta / split_fastq (src/analysis.c:2372-2387,2584-2596,2440-2466) on the reader fixtures: CRLF,
    multi-line, blank lines, bad quality, FASTA, no final newline -- including the comment a record inherits from the one
    before it"""
    p = subprocess.run([CLI, "analysis", args[0], fq] + args[1:], cwd=KSEQ, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0
    err = b"".join(l for l in p.stderr.splitlines(True) if b"read number:" in l)
    assert p.stdout + b"--- stderr ---\n" + err == open(os.path.join(ANA, "%s.%s.txt" % (fq, tag)), "rb").read()


def test_analysis_usage_and_errors(built, tmp_path):
    p = subprocess.run([CLI, "analysis"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"analysis ana_meta" in p.stderr and p.stdout == b""
    p = subprocess.run([CLI, "analysis", "nope"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert b"command [nope] unsupported!" in p.stderr
    (tmp_path / "empty.sam").write_bytes(b"@HD\tVN:1.0\n")
    shutil.copy(os.path.join(ANA, "nodes.dmp"), tmp_path / "nodes.dmp")
    p = subprocess.run([CLI, "analysis", "ana_meta", "empty.sam", "nodes.dmp"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == b"Current read empty.sam.temp\tempty.sam.temp\t"       # (no record: the reference aborts in skip_sam_head, src/analysis.c:338-351)
