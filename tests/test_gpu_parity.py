"""GPU parity tests proper: everything goes through the C-ABI of libdesamba_amd.so on a real MI355X and is
compared bit for bit with the oracle / the committed golden SAM."""
import hashlib
import os

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(demo):
    import desamba_amd as D
    idx = D.Index(demo["index"])
    ctx = D.Ctx(idx, 0)
    yield D, idx, ctx
    ctx.close(); idx.close()


def classify_all(D, ctx, recs, chunk=None):
    """-> (list of hit-key lists, SAM bytes); history is reset like a new input file"""
    ctx.reset_history()
    out, sam = [], []
    chunk = chunk or len(recs) or 1
    for s in range(0, max(len(recs), 1), chunk):
        part = recs[s:s + chunk]
        reads = D.make_reads(part)
        res = ctx.classify(reads)
        for i in range(len(part)):
            rr = res.reads[i]
            assert rr.status == 0
            out.append([res.hits[rr.first + k].key() for k in range(rr.n)])
        sam.append(ctx.sam(res))
    return out, b"".join(sam)


def test_demo_sam_md5(gpu, demo, golden_md5):
    """config 1: the reference's quick-start run, byte-identical SAM"""
    D, idx, ctx = gpu
    hits, sam = classify_all(D, ctx, D.read_fastq(demo["fastq"]))
    assert hashlib.md5(sam).hexdigest() == golden_md5


@pytest.mark.parametrize("name", ["ont20k", "ngs150", "pb", "ont5k_e25", "appc", "heavy", "wrapq"])
def test_synthetic_golden_sam(gpu, name):
    D, idx, ctx = gpu
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
    assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()


@pytest.mark.parametrize("name", ["heavy", "ont5k_e25", "wrapq"])
def test_heavy_first_launch(gpu, name, monkeypatch):
    """the early launch of the heaviest reads (second stream, own slots) must not change any result"""
    D, idx, ctx = gpu
    monkeypatch.setenv("DSB_HEAVY_FIRST", "16")
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
    assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()


@pytest.mark.parametrize("name", ["heavy", "ont20k", "pb"])
def test_second_run_with_large_node_arena(gpu, name, monkeypatch):
    """reads that overflow the match-node arena of their slot (a kvec in the reference, src/cly.c:2532-2819) are
    run again in the large-arena slots; a tiny first-level arena forces that path"""
    D, idx, ctx = gpu
    monkeypatch.setenv("DSB_SMS_CAP", "96")
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
    assert ctx.timing().n_retry > 0
    assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()
    monkeypatch.delenv("DSB_SMS_CAP")
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
    assert ctx.timing().n_retry == 0
    assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()


def test_rank64_layout_on_device(demo, monkeypatch):
    """the 64-bit superblock rank layout (indexes beyond 2^32 BWT symbols), forced on for the demo index"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_FORCE_RANK64", "1")
    idx = D.Index(demo["index"]); ctx = D.Ctx(idx, 0)
    try:
        for name in ("ont20k", "pb", "ngs150"):
            hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
            assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read(), name
    finally:
        ctx.close(); idx.close()


def test_stage_parity_seed_lookup(gpu, demo, oracle):
    """exist-kmer bits of every window and the seed lists (a-3) of both strands"""
    D, idx, ctx = gpu
    recs = D.read_fastq(demo["fastq"], 64) + D.read_fastq(os.path.join(GOLDEN, "synth", "ngs150.fq"), 64)
    reads = D.make_reads(recs)
    ctx.reset_history(); ctx.classify(reads)
    hist = 0
    for i, (nm, seq, q) in enumerate(recs):
        oracle.classify(seq, hist); hist = max(hist, len(seq))
        n = len(seq) - 16 + 1
        for s in (1, 0):
            assert ctx.exist_bits(i, s) == bytes(oracle.exist_bits(seq, s)[:n])
            assert ctx.seeds(i, s) == oracle.seeds(s)


def test_long_reads_vs_oracle(gpu, demo, oracle, tmp_path):
    """full-size reads (50 kbp ONT-15%), fresh seed, checked hit by hit against the oracle"""
    import subprocess
    D, idx, ctx = gpu
    fq = tmp_path / "ont50k.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "1536", "50000", "0.15", "4242", "ont"])
    recs = D.read_fastq(str(fq))
    hits, _ = classify_all(D, ctx, recs)
    hist = 0
    for (nm, seq, q), got in zip(recs, hits):
        assert got == oracle.classify(seq, hist), nm
        hist = max(hist, len(seq))


def test_edge_cases(gpu, oracle):
    D, idx, ctx = gpu
    recs = [(b"short", b"ACGT" * 9, None), (b"min", b"ACGTTGCA" * 5, None), (b"polyA", b"A" * 300, None),
            (b"allN", b"N" * 200, None), (b"lower", b"acgtnnacgt" * 30, None), (b"l39", b"A" * 39, None), (b"empty", b"", None)]
    hits, sam = classify_all(D, ctx, recs)
    hist = 0
    for (nm, seq, q), got in zip(recs, hits):
        assert got == oracle.classify(seq, hist), nm
        hist = max(hist, len(seq))
    assert sam.count(b"\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\t\n") >= 4
    # empty batch
    res = ctx.classify(D.make_reads([]))
    assert res.n_hits == 0


def test_batch_split_invariance(gpu, demo):
    """results do not depend on how the input is cut into batches or on the number of reads in flight:
    the running max_read_l (oracle U4) is carried across batches"""
    D, idx, ctx = gpu
    recs = D.read_fastq(os.path.join(GOLDEN, "synth", "pb.fq")) + D.read_fastq(os.path.join(GOLDEN, "synth", "ngs150.fq"), 100)
    whole, sam_whole = classify_all(D, ctx, recs)
    split, sam_split = classify_all(D, ctx, recs, chunk=7)
    assert whole == split and sam_whole == sam_split
    ctx2 = D.Ctx(idx, 0, n_slots=3)
    few, sam_few = classify_all(D, ctx2, recs)
    ctx2.close()
    assert few == whole


def test_idempotent_rerun(gpu, demo):
    D, idx, ctx = gpu
    recs = D.read_fastq(demo["fastq"], 200)
    reads = D.make_reads(recs)
    ctx.reset_history(); ctx.upload(reads)
    ctx.run(); a = ctx.sam(ctx.fetch())
    ctx.run(); b = ctx.sam(ctx.fetch())
    assert a == b


def test_cli_drop_in(gpu, demo, golden_md5, tmp_path):
    """the `deSAMBA classify` CLI: same command line, byte-identical output file"""
    import subprocess
    out = tmp_path / "cli.sam"
    cli = os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA")
    p = subprocess.run([cli, "classify", "-t", "4", demo["index"], demo["fastq"], "-o", str(out)], stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert b"1237 sequences processed in" in p.stderr and b"loading index" in p.stderr
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == golden_md5


@pytest.mark.parametrize("args,fq,exp", [(["-f", "SAM_FULL"], "ngs150.fq", "ngs150.full.ubfree.sam"),
                                         (["-l", "100", "-s", "30", "-r", "2"], "pb.fq", "pb.l100s30r2.ubfree.sam"),
                                         (["-f", "DES"], "pb.fq", "pb.des.ubfree.txt"),
                                         (["-f", "DES_FULL", "-r", "1"], "ngs150.fq", "ngs150.desfull.ubfree.txt"),
                                         (["-f", "DES", "-r", "1"], "ont5k_e25.fq", "ont5k_e25.des_r1.ubfree.txt")])
def test_cli_options(gpu, tmp_path, args, fq, exp):
    """SAM_FULL (SEQ and QUAL columns), DES / DES_FULL (n_rst, n_anc, FAST/SLOW) and non-default -l/-s/-r, against the
    reference's output for the same command line"""
    import subprocess
    from conftest import ROOT as R
    out = tmp_path / "out.sam"
    subprocess.check_call([os.path.join(R, "desamba_amd", "bin", "deSAMBA"), "classify"] + args +
                          [os.path.join(R, "data", "demo", "index"), os.path.join(GOLDEN, "synth", fq), "-o", str(out)], stderr=subprocess.DEVNULL)
    assert out.read_bytes() == open(os.path.join(GOLDEN, "synth", exp), "rb").read()


def test_cli_many_batches_files_and_gzip(gpu, tmp_path, monkeypatch):
    """the CLI pipeline with buffers of 256 KB (dozens of batches alternating between the two device contexts, records
    carried over buffer ends), several input files (history restarts per file) and gzip input"""
    import gzip
    import subprocess
    names = ["ont20k", "ngs150", "pb", "appc", "wrapq"]
    exp = b"".join(open(os.path.join(GOLDEN, "synth", n + ".ubfree.sam"), "rb").read() for n in names)
    files = []
    for i, n in enumerate(names):
        src = os.path.join(GOLDEN, "synth", n + ".fq")
        if i % 2:
            dst = tmp_path / (n + ".fq.gz")
            with gzip.open(dst, "wb") as f:
                f.write(open(src, "rb").read())
            files.append(str(dst))
        else:
            files.append(src)
    monkeypatch.setenv("DSB_CLI_BATCH_KB", "256")
    out = tmp_path / "out.sam"
    from conftest import ROOT as R
    subprocess.check_call([os.path.join(R, "desamba_amd", "bin", "deSAMBA"), "classify", os.path.join(R, "data", "demo", "index")] + files + ["-o", str(out)],
                          stderr=subprocess.DEVNULL)
    assert out.read_bytes() == exp


def test_full_size_batch_properties(gpu, demo, oracle, tmp_path):
    """BASELINE-size reads (50 kbp) in a batch large enough to exercise the work queue, the work ordering and
    many waves per CU: (1) a shuffled copy of the batch gives the same per-read hits (order independence),
    (2) two different batch splits give identical SAM (checksum of checksums), (3) a sample is checked hit by
    hit against the oracle, (4) every read of this error profile is classified to its source reference."""
    import hashlib, random, subprocess
    D, idx, ctx = gpu
    fq = tmp_path / "full.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "1536", "50000", "0.15", "31337", "ont"])
    recs = D.read_fastq(str(fq))
    whole, sam_whole = classify_all(D, ctx, recs)
    split, sam_split = classify_all(D, ctx, recs, chunk=500)
    assert hashlib.md5(sam_whole).hexdigest() == hashlib.md5(sam_split).hexdigest()
    perm = list(range(len(recs))); random.Random(7).shuffle(perm)
    shuf, _ = classify_all(D, ctx, [recs[i] for i in perm])
    for k, i in enumerate(perm):
        assert shuf[k] == whole[i]
    for i in random.Random(11).sample(range(len(recs)), 48):
        assert whole[i] == oracle.classify(recs[i][1], 50000), recs[i][0]
    mapped = sum(1 for h in whole if h)
    assert mapped == len(recs)
    # read names carry the truth (r{i}_{refIndex}_{start}_{F|R}): the primary hit is on the source reference
    ok = sum(1 for (nm, s, q), h in zip(recs, whole) if h and h[0][0] == int(nm.split(b"_")[1]))
    assert ok >= 0.99 * len(recs)


def test_upload_fastq_equals_upload(gpu, demo):
    """dsb_batch_upload_fastq (the library parses the file) == dsb_batch_upload of the same records"""
    D, idx, ctx = gpu
    recs = D.read_fastq(demo["fastq"], 300)
    ctx.reset_history(); a = ctx.sam(ctx.classify(D.make_reads(recs)))
    ctx.reset_history()
    n = ctx.upload_fastq(demo["fastq"], 0, 300)
    assert n == 300
    ctx.reads = D.make_reads(recs)          # names for the SAM formatter only
    ctx.run(); b = ctx.sam(ctx.fetch())
    assert a == b
    assert ctx.upload_fastq(demo["fastq"], 1200, 1000) == 37


def test_second_index_golden(strain):
    """a second index (synthetic strains + tandem repeats, 4 Mbp): byte-identical SAM against the reference's, with
    reads whose match-node list outgrows their slot's arena going through the second run by themselves"""
    import desamba_amd as D
    idx = D.Index(strain["index"]); ctx = D.Ctx(idx, 0)
    try:
        hits, sam = classify_all(D, ctx, D.read_fastq(strain["fastq"]))
        assert ctx.timing().n_retry > 0
        assert sam == open(strain["sam"], "rb").read()
    finally:
        ctx.close(); idx.close()
