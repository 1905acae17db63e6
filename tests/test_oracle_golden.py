"""The oracle (oracle/*.c) against the reference's own outputs: the demo quick-start SAM (md5),
golden SAM of synthetic sets written by the reference's UB-pinned build, and SURVEY.md Appendix C."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT, md5_file, sam_lines

SETS = ["ont20k", "ngs150", "pb", "ont5k_e25", "appc", "heavy", "wrapq", "ngs_e14", "overhang", "manyanchors"]


def test_demo_md5(demo, oracle, golden_md5, tmp_path):
    out = tmp_path / "demo.sam"
    n, bases = oracle.classify_file(demo["fastq"], str(out), threads=4)
    assert n == 1237 and bases == 2153923
    assert md5_file(str(out)) == golden_md5 == "1da908b61be240c40334b58d3c12ba2a"
    assert sam_lines(str(out))[:60] == sam_lines(os.path.join(GOLDEN, "demo_head60.sam"))


@pytest.mark.parametrize("name", SETS)
def test_synthetic_golden(demo, oracle, name, tmp_path):
    """bit-exact against the reference built with its UB pinned (oracle.h U1-U5)"""
    out = tmp_path / (name + ".sam")
    oracle.classify_file(os.path.join(GOLDEN, "synth", name + ".fq"), str(out), threads=2)
    assert sam_lines(str(out)) == sam_lines(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"))


def test_ultralong_read(demo, oracle, tmp_path):
    """a read of 0.87 Mbp: beyond the 786432 bases from which the stock reference writes behind its 9-mer table
    (2^20 nodes for 2^18 + L, src/cly_mt.c:540-541, src/cly.c:2173-2224); the golden SAM is the UB-pinned build's, whose
    table holds 2^24 nodes (oracle/Makefile U7).  Behaviour of this build: such reads are classified like any other"""
    import gzip
    fq = tmp_path / "ultralong.fq"
    fq.write_bytes(gzip.open(os.path.join(GOLDEN, "synth", "ultralong.fq.gz")).read())
    out = tmp_path / "ultralong.sam"
    oracle.classify_file(str(fq), str(out), threads=1)
    assert sam_lines(str(out)) == sam_lines(os.path.join(GOLDEN, "synth", "ultralong.ubfree.sam"))


# (no stock output for `overhang`: the stock binary crashes on it, oracle.h U6)
@pytest.mark.parametrize("name", [n for n in SETS if n not in ("ngs_e14", "overhang")])
def test_distance_to_stock_reference(name):
    """the stock binary differs from the UB-free build only in AS / POS / CIGAR digits, never in the
    per-read list of (flag, reference) -- SURVEY.md 8a-UB"""
    a = sam_lines(os.path.join(GOLDEN, "synth", name + ".stock.sam"))
    b = sam_lines(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"))
    assert len(a) == len(b)
    for x, y in zip(a, b):
        fx, fy = x.split(b"\t"), y.split(b"\t")
        assert fx[:3] == fy[:3]


def test_history_runs_over_all_input_files(demo, oracle, tmp_path):
    """max_read_l is never reset between input files (src/cly_mt.c:538-556, src/cly.c:2958): the golden SAM of ONE reference
    run over six files (150-bp reads behind a 20-kbp file are filtered in 3G mode) equals the oracle's for their
    concatenation, and differs from the concatenation of six single-file runs"""
    names = ["ont20k", "ngs_e14", "pb", "appc", "wrapq", "ngs150"]
    cat = tmp_path / "cat.fq"
    cat.write_bytes(b"".join(open(os.path.join(GOLDEN, "synth", n + ".fq"), "rb").read() for n in names))
    out = tmp_path / "cat.sam"
    oracle.classify_file(str(cat), str(out), threads=2)
    exp = open(os.path.join(GOLDEN, "synth", "multi6.ubfree.sam"), "rb").read()
    assert out.read_bytes() == exp
    assert exp != b"".join(open(os.path.join(GOLDEN, "synth", n + ".ubfree.sam"), "rb").read() for n in names)


def test_appendix_c_history_independence(demo, oracle):
    """r57290 must score AS:i:125 whether or not r57271 precedes it (the stock reference gives 110 then)"""
    import desamba_amd as D
    recs = D.read_fastq(os.path.join(GOLDEN, "synth", "appc.fq"))
    alone = oracle.classify(recs[1][1], 0)
    oracle.classify(recs[0][1], 0)
    after = oracle.classify(recs[1][1], len(recs[0][1]))
    assert alone == after
    assert alone[0][5] == 125


def test_thread_invariance(demo, oracle, tmp_path):
    a, b = tmp_path / "t1.sam", tmp_path / "t4.sam"
    fq = os.path.join(GOLDEN, "synth", "pb.fq")
    oracle.classify_file(fq, str(a), threads=1); oracle.classify_file(fq, str(b), threads=4)
    assert md5_file(str(a)) == md5_file(str(b))


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "deSAMBA_ubfree")), reason="reference build absent")
def test_against_live_reference(demo, oracle, tmp_path):
    """run the compiled reference (UB-pinned build) here and compare on a fresh synthetic set"""
    fq = tmp_path / "live.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "60", "8000", "0.15", "77", "ont"])
    ref_out, ora_out = tmp_path / "ref.sam", tmp_path / "ora.sam"
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "deSAMBA_ubfree"), "classify", "-t", "2", demo["index"], str(fq), "-o", str(ref_out)],
                          stderr=subprocess.DEVNULL)
    oracle.classify_file(str(fq), str(ora_out), threads=2)
    assert sam_lines(str(ora_out)) == sam_lines(str(ref_out))


def test_second_index_golden(strain, tmp_path):
    """the oracle on a second index (synthetic strains + tandem repeats: several near-equal targets per read, match-node
    lists of tens of thousands of nodes) against the reference's SAM for the same reads"""
    import oracle_lib
    ora = oracle_lib.Oracle(strain["index"])
    out = str(tmp_path / "strain.sam")
    ora.classify_file(strain["fastq"], out)
    assert open(out, "rb").read() == open(strain["sam"], "rb").read()


KSEQ = ["four.fq", "crlf.fq", "multi.fq", "blank.fq", "badqual.fq", "no_nl.fq"]


def _unmapped_full(recs):
    return b"".join(n + b"\t4\t*\t0\t0\t*\t*\t0\t0\t" + s + b"\t" + (q if q is not None else b"(null)") + b"\t\n" for n, s, q in recs)


@pytest.mark.parametrize("name", KSEQ)
def test_kseq_rules_against_reference(demo, oracle, name, tmp_path):
    """tests/golden/kseq: awkward texts of reads shorter than 40 bases (unmapped, so SAM_FULL shows name / sequence /
    quality as parsed) with the output of the reference binary: pins the three restatements of the reference's
    kseq_read (the oracle's reader, the Python reader of the test helpers, and tests/test_cli_parser.py's
    character-level one that the CLI's in-place parser is tested against)"""
    import desamba_amd as D
    import test_cli_parser as T
    path = os.path.join(GOLDEN, "kseq", name)
    exp = open(path + ".full.ref.sam", "rb").read()
    data = open(path, "rb").read()
    assert _unmapped_full(T.kseq_records(data)) == exp
    assert _unmapped_full(D.parse_fastq(data)) == exp
    out = tmp_path / "o.sam"
    oracle.classify_file(path, str(out), full=1)
    assert out.read_bytes() == exp


def test_fasta_every_record_is_classified(demo, oracle, tmp_path):
    """FASTA input: the reference loses every other record on the first use of a kseq_t slot (its look-ahead character
    lives in the slot, the stream is shared: src/cly_mt.c:42-56, src/lib/utils.c:939-945) -- tests/golden/kseq/records.fa
    shows it: c0, c2, c4, c6 of eight.  This build classifies every record (documented deviation); the records
    the reference does keep are byte-identical."""
    import desamba_amd as D
    path = os.path.join(GOLDEN, "kseq", "records.fa")
    ref = open(path + ".full.ref.sam", "rb").read().splitlines(True)
    recs = D.parse_fastq(open(path, "rb").read())
    assert [r[0] for r in recs] == [b"c%d" % i for i in range(8)]
    ours = _unmapped_full(recs).splitlines(True)
    assert ref == ours[0::2]
    out = tmp_path / "o.sam"
    oracle.classify_file(path, str(out), full=1)
    assert out.read_bytes().splitlines(True) == ours
