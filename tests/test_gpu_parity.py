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


@pytest.fixture(autouse=True)
def _switches_of_the_shared_ctx(request):
    """the DSB_* switches are read once per ctx: after a test that set some (monkeypatch has put the environment back by now) the
    shared ctx reads them again, so that the next test starts from the defaults"""
    yield
    if "gpu" in request.fixturenames:
        request.getfixturevalue("gpu")[2].reload_env()


def classify_all(D, ctx, recs, chunk=None):
    """-> (list of hit-key lists, SAM bytes); history is reset like a new input file"""
    ctx.reset_history(); ctx.reload_env()          # (the DSB_* switches a test sets are read once per ctx: read them again)
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


@pytest.mark.parametrize("name", ["ont20k", "ngs150", "pb", "ont5k_e25", "appc", "heavy", "wrapq", "ngs_e14", "overhang", "manyanchors"])
def test_synthetic_golden_sam(gpu, name):
    D, idx, ctx = gpu
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
    assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()


def test_ultralong_read(gpu, tmp_path):
    """one read of 0.87 Mbp (longer than the 786432 bases from which the stock reference overruns its 9-mer table,
    src/cly_mt.c:540-541): through the library and through the CLI from the .gz file, byte-identical to the UB-pinned
    reference whose table is large enough (oracle/Makefile U7)"""
    import gzip
    import subprocess
    D, idx, ctx = gpu
    gz = os.path.join(GOLDEN, "synth", "ultralong.fq.gz")
    exp = open(os.path.join(GOLDEN, "synth", "ultralong.ubfree.sam"), "rb").read()
    hits, sam = classify_all(D, ctx, D.parse_fastq(gzip.open(gz).read()))
    assert sam == exp
    out = tmp_path / "ul.sam"
    subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify", os.path.join(ROOT, "data", "demo", "index"), gz, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    assert out.read_bytes() == exp


def test_cli_fasta_compat_switch(gpu, tmp_path, monkeypatch):
    """DSB_FASTA_COMPAT=1: the CLI loses the FASTA records the reference loses (every other one on the first use of a
    kseq_t slot): SAM_FULL == the reference binary's own output for tests/golden/kseq/records.fa"""
    import subprocess
    path = os.path.join(GOLDEN, "kseq", "records.fa")
    monkeypatch.setenv("DSB_FASTA_COMPAT", "1")
    out = tmp_path / "fa.sam"
    subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify", "-f", "SAM_FULL", os.path.join(ROOT, "data", "demo", "index"), path, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    assert out.read_bytes() == open(path + ".full.ref.sam", "rb").read()


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
    ctx.reset_history(); ctx.reload_env(); ctx.classify(reads)
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
    carried over buffer ends), several input files and gzip input.  max_read_l runs over ALL files, as in the reference
    (its per-thread buffers are allocated once, before the loop over the files: src/cly_mt.c:538-556): the expected
    output is ONE reference run over the six files, in which the 150-bp reads behind the 20-kbp file are filtered in 3G
    mode -- not the concatenation of six single-file runs"""
    import gzip
    import subprocess
    names = ["ont20k", "ngs_e14", "pb", "appc", "wrapq", "ngs150"]
    exp = open(os.path.join(GOLDEN, "synth", "multi6.ubfree.sam"), "rb").read()
    assert exp != b"".join(open(os.path.join(GOLDEN, "synth", n + ".ubfree.sam"), "rb").read() for n in names)
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


def _cli(args, out):
    import subprocess
    p = subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify"] + args + ["-o", str(out)], stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    return p.stderr


def test_cli_two_logical_shards_on_one_device(gpu, tmp_path, monkeypatch):
    """`-g 0,0`: the multi-GPU host path (one worker thread and two contexts per listed device, batches dealt to whichever
    is free, max_read_l carried in the batch header, ordered writer) with two logical shards on the one device of the
    test box; expected = ONE reference run over the four files"""
    monkeypatch.setenv("DSB_CLI_BATCH_KB", "128")
    files = [os.path.join(GOLDEN, "synth", n + ".fq") for n in ("pb", "ngs_e14", "ngs150", "appc")]
    out = tmp_path / "out.sam"
    _cli(["-g", "0,0", os.path.join(ROOT, "data", "demo", "index")] + files, out)
    assert out.read_bytes() == open(os.path.join(GOLDEN, "synth", "multi4.ubfree.sam"), "rb").read()
    out2 = tmp_path / "out2.sam"
    _cli(["-g", "all", os.path.join(ROOT, "data", "demo", "index")] + files, out2)
    assert out2.read_bytes() == out.read_bytes()


def test_multi_api_cuts_a_batch_into_device_filling_calls(demo, tmp_path):
    """dsb_multi_classify_batch on 131072 x 5 kbp reads over two contexts of device 0: the batch is cut into n / W reads per call
    (64 k here), not into 64-Mbase crumbs -- at most two dsb_classify_batch calls per context, the same hits as one context on the whole
    batch, and a rate within 10 % of it (round 3's 64-Mbase chunks ran ~13 k such reads per call)"""
    import subprocess, time
    import desamba_amd as D
    fq = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path), "dsb_multi_test.fq")
    subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "gen_fastq.py"), demo["index"], fq, "131072", "5000", "0.15", "77", "ont", "8"], check=True, stdout=subprocess.DEVNULL)
    idx = D.Index(demo["index"]); ctx = None; m = None
    try:
        lines = open(fq, "rb").read().split(b"\n"); os.remove(fq)          # (four-line records: the general parser takes minutes on 1.3 GB)
        recs = [(lines[i][1:], lines[i + 1], None) for i in range(0, len(lines) - 3, 4)]
        del lines
        assert len(recs) == 131072
        reads = D.make_reads(recs)
        ctx = D.Ctx(idx, 0, max_read_len=5064, max_batch_reads=131072, max_batch_bases=131072 * 5100)
        ctx.classify(reads)                                  # (buffers, arenas: grown once)
        t0 = time.perf_counter(); res1 = ctx.classify(reads); t_single = time.perf_counter() - t0
        one = [[res1.hits[res1.reads[i].first + k].key() for k in range(res1.reads[i].n)] for i in range(0, len(recs), 37)]
        ctx.close(); ctx = None
        m = D.Multi(idx, [0, 0])
        m.classify(reads); m.reset_history()
        t0 = time.perf_counter(); res2 = m.classify(reads); t_multi = time.perf_counter() - t0
        calls = m.last_calls()
        assert sum(calls) == 2 and max(calls) <= 2, calls
        two = [[res2.hits[res2.reads[i].first + k].key() for k in range(res2.reads[i].n)] for i in range(0, len(recs), 37)]
        assert two == one
        print("131072 x 5 kbp: one context %.3f s, dsb_multi_classify_batch over two contexts of one device %.3f s, calls per context %r" % (t_single, t_multi, calls))
        # measured: 0.25-0.29 s against 0.24 s -- each half ends in the tail of its own heaviest reads (tandem repeats of the demo index), which
        # only partly hides behind the other half's main launch; round 3's 64-Mbase chunks (10 calls of 13 k reads) took 2x
        assert t_multi <= 1.5 * t_single
    finally:
        if ctx:
            ctx.close()
        if m:
            m.close()
        idx.close()


def test_multi_api_shards_a_batch(gpu, monkeypatch):
    """dsb_ctx_create_multi + dsb_multi_classify_batch: one batch cut by dsb_shard_plan into chunks of 50 reads over two
    contexts (device 0 listed twice): same hits as the single context, in input order, history carried per chunk"""
    D, idx, ctx = gpu
    recs = D.read_fastq(os.path.join(GOLDEN, "synth", "pb.fq")) + D.read_fastq(os.path.join(GOLDEN, "synth", "ngs_e14.fq")) + \
        D.read_fastq(os.path.join(GOLDEN, "synth", "ngs150.fq"), 150)
    single, sam_single = classify_all(D, ctx, recs)
    monkeypatch.setenv("DSB_SHARD_CHUNK_READS", "50")
    m = D.Multi(idx, [0, 0])
    try:
        reads = D.make_reads(recs)
        res = m.classify(reads)
        got = [[res.hits[res.reads[i].first + k].key() for k in range(res.reads[i].n)] for i in range(len(recs))]
        assert got == single
        assert D.format_sam(idx, reads, res) == sam_single
        # a second batch continues the history of the first (150-bp reads stay in 3G mode)
        tail = D.read_fastq(os.path.join(GOLDEN, "synth", "ngs_e14.fq"))
        res2 = m.classify(D.make_reads(tail))
        ctx.reset_history(); ctx.classify(D.make_reads(recs))
        exp2 = ctx.classify(D.make_reads(tail))
        assert [[res2.hits[res2.reads[i].first + k].key() for k in range(res2.reads[i].n)] for i in range(len(tail))] == \
            [[exp2.hits[exp2.reads[i].first + k].key() for k in range(exp2.reads[i].n)] for i in range(len(tail))]
    finally:
        m.close()


@pytest.mark.parametrize("knob,value,counter", [("DSB_HOUT_CAP", "8", "n_regrow"), ("DSB_STEP_LIMIT_RT", "300", "n_retry"), ("DSB_ANC_CAP_RT", "64", "n_retry")])
def test_capacity_overruns_are_rerun_not_fatal(gpu, monkeypatch, knob, value, counter):
    """the reference's per-read lists are unbounded and it has no loop budget; here a full hit buffer is regrown and the
    reads that found it full run again, and reads that outgrow the anchor array or the loop budget of their wave slot run
    again in the large second-run slots (8x anchors, 16x budget) -- forced by tiny capacities; results unchanged"""
    D, idx, ctx = gpu
    for name in ("ont20k", "pb", "ngs150"):
        monkeypatch.setenv(knob, value)
        hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
        if name != "ngs150" or knob == "DSB_HOUT_CAP":      # (150-bp reads stay below the tiny anchor array and loop budget)
            assert getattr(ctx.timing(), counter) > 0, (name, knob)
        assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read(), (name, knob)
        monkeypatch.delenv(knob)
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", "pb.fq")))
    assert ctx.timing().n_retry == 0 and ctx.timing().n_regrow == 0


def test_input_slots_keep_batches_resident(demo):
    """a ctx with several input slots holds several staged batches in HBM; they can be run in any order, repeatedly"""
    import desamba_amd as D
    idx = D.Index(demo["index"]); ctx = D.Ctx(idx, 0, input_slots=3)
    try:
        names = ["pb", "ont20k", "ngs150"]
        reads = []
        for k, n in enumerate(names):
            ctx.select_slot(k); ctx.set_history(0)
            reads.append(D.make_reads(D.read_fastq(os.path.join(GOLDEN, "synth", n + ".fq"))))
            ctx.upload(reads[k])
        for k in (2, 0, 1, 0, 2):
            ctx.select_slot(k); ctx.run()
            assert ctx.sam(ctx.fetch(), reads=reads[k]) == open(os.path.join(GOLDEN, "synth", names[k] + ".ubfree.sam"), "rb").read(), names[k]
        with pytest.raises(D.DsbError):
            ctx.select_slot(3)
    finally:
        ctx.close(); idx.close()


@pytest.mark.parametrize("name", ["four.fq", "crlf.fq", "multi.fq", "blank.fq", "badqual.fq", "no_nl.fq"])
def test_cli_reads_what_the_reference_reads(gpu, tmp_path, name, monkeypatch):
    """the record rules of the reference's kseq_read ('\\r' kept, empty lines inside a sequence, whole-line quality, records
    with a quality string of the wrong length dropped): SAM_FULL of the CLI == SAM_FULL of the reference binary
    (tests/golden/kseq), also with a buffer barely larger than a record"""
    path = os.path.join(GOLDEN, "kseq", name)
    exp = open(path + ".full.ref.sam", "rb").read()
    for kb in (None, "64"):
        if kb:
            monkeypatch.setenv("DSB_CLI_BATCH_KB", kb)
        out = tmp_path / "o.sam"
        _cli(["-f", "SAM_FULL", os.path.join(ROOT, "data", "demo", "index"), path], out)
        assert out.read_bytes() == exp


def test_cli_fasta_classifies_every_record(gpu, tmp_path):
    """FASTA: the reference loses every other record (tests/golden/kseq/records.fa.full.ref.sam: c0, c2, c4, c6 of eight);
    this CLI classifies all eight, and the records the reference keeps are byte-identical (documented deviation)"""
    path = os.path.join(GOLDEN, "kseq", "records.fa")
    ref = open(path + ".full.ref.sam", "rb").read().splitlines(True)
    out = tmp_path / "o.sam"
    _cli(["-f", "SAM_FULL", os.path.join(ROOT, "data", "demo", "index"), path], out)
    ours = out.read_bytes().splitlines(True)
    assert len(ours) == 8 and ours[0::2] == ref


def test_device_work_counters_match_the_oracle(gpu, demo, oracle, tmp_path):
    """dsb_timing.n_occ / n_mem / n_sa / ref_bases (the terms of the classify kernels' algorithmic bytes, counted on the
    device) against the oracle's counters on the same reads: within 1 % (the device also walks the few islands that the
    reference skips after a score > 512)"""
    import subprocess
    D, idx, ctx = gpu
    fq = tmp_path / "c.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "1024", "50000", "0.15", "99", "ont"])
    recs = D.read_fastq(str(fq))
    classify_all(D, ctx, recs)
    t = ctx.timing()
    tot = [0, 0, 0, 0]
    for (nm, seq, q) in recs:
        oracle.classify(seq, 50000)
        oc = oracle.counters()
        for k, j in enumerate((2, 5, 3, 4)):
            tot[k] += oc[j]
    got = [t.n_occ, t.n_mem, t.n_sa, t.ref_bases]
    for a, b in zip(got, tot):
        assert b <= a <= 1.01 * b, (got, tot)
    assert t.main_occ <= t.n_occ


@pytest.mark.parametrize("name", ["ont20k", "ngs150", "pb", "ont5k_e25", "appc", "heavy", "wrapq", "ngs_e14", "overhang"])
def test_seed_scan_kernel_golden_sam(gpu, name, monkeypatch):
    """k_seed_scan (one lane per read strand, probing only the windows the reference's scan consumes, seed lists written by
    the kernel) forced on for small batches too -- by default it serves batches of >= 2048 reads"""
    D, idx, ctx = gpu
    monkeypatch.setenv("DSB_SEED_SCAN", "1")
    hits, sam = classify_all(D, ctx, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))
    assert ctx.timing().seed_scan == 1
    assert sam == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()


def test_seed_scan_kernel_stage_parity(gpu, demo, oracle, monkeypatch):
    """the seed lists k_seed_scan writes (both strands, top marks, total_score) against the oracle's, read for read; the
    demo set is ragged (reads of 300 .. 4000 bases share wavefronts, longest first) and includes reads shorter than 40"""
    D, idx, ctx = gpu
    monkeypatch.setenv("DSB_SEED_SCAN", "1")
    recs = D.read_fastq(demo["fastq"], 300) + D.read_fastq(os.path.join(GOLDEN, "synth", "ngs150.fq"), 64) + [(b"short", b"ACGT" * 9, None), (b"polyA", b"A" * 300, None)] + \
        D.read_fastq(os.path.join(GOLDEN, "synth", "ont20k.fq"), 4)
    reads = D.make_reads(recs)
    ctx.reset_history(); ctx.reload_env(); ctx.classify(reads)
    t = ctx.timing()
    assert t.seed_scan == 1 and 0 < t.windows < 1.2 * t.bases and t.probes_t1 < t.windows
    hist = 0
    for i, (nm, seq, q) in enumerate(recs):
        oracle.classify(seq, hist); hist = max(hist, len(seq))
        for s in (1, 0):
            assert ctx.seeds(i, s) == oracle.seeds(s), (nm, s)
    # the hit bits are still available as a stage dump (probed on demand)
    n = len(recs[0][1]) - 16 + 1
    assert ctx.exist_bits(0, 1) == bytes(oracle.exist_bits(recs[0][1], 1)[:n])


def test_seed_scan_kernel_long_reads_vs_oracle(gpu, demo, oracle, tmp_path, monkeypatch):
    """a batch large enough for the default choice (>= 2048 reads): 2304 x 50 kbp through k_seed_scan + k_classify, a sample
    checked hit by hit against the oracle, every read classified to its source reference, device work counters sane"""
    import random, subprocess
    D, idx, ctx = gpu
    monkeypatch.delenv("DSB_SEED_SCAN", raising=False)
    fq = tmp_path / "scan.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "2304", "50000", "0.15", "555", "ont"])
    recs = D.read_fastq(str(fq))
    hits, _ = classify_all(D, ctx, recs)
    t = ctx.timing()
    assert t.seed_scan == 1
    assert 0.7 * t.bases < t.windows < 1.05 * t.bases          # ~0.9 probes per base (all windows of both strands: 2.0)
    for i in random.Random(3).sample(range(len(recs)), 64):
        assert hits[i] == oracle.classify(recs[i][1], 50000), recs[i][0]
    ok = sum(1 for (nm, s, q), h in zip(recs, hits) if h and h[0][0] == int(nm.split(b"_")[1]))
    assert ok >= 0.99 * len(recs)


def _hash64_1(key):
    M = (1 << 64) - 1
    key = (~key + (key << 21)) & M; key ^= key >> 24; key = (key + (key << 3) + (key << 8)) & M
    key ^= key >> 14; key = (key + (key << 2) + (key << 4)) & M; key ^= key >> 28; key = (key + (key << 31)) & M
    return key


def _hash64_2(key):
    M = (1 << 64) - 1
    key = (key + (~(key << 32) & M)) & M; key ^= key >> 22; key = (key + (~(key << 13) & M)) & M; key ^= key >> 8
    key = (key + (key << 3)) & M; key ^= key >> 15; key = (key + (~(key << 27) & M)) & M; key ^= key >> 31
    return key


def test_seed_lookup_on_synthetic_multi_gib_tables(demo, tmp_path, monkeypatch):
    """the measurement hook for the HBM regime of the seed lookup: 2 x 2 GiB synthetic exist-kmer tables
    (k = 18, 34-bit mask, 20 % of the bits set; dsb_ctx_use_synthetic_filter).  Both seed-lookup kernels must answer what
    a host recomputation of get_exist_kmer (src/cly.c:956-972: hash64_1 / hash64_2, low-complexity filter of store_kmers)
    answers on those tables: every window of a sample of reads (all-windows probe), and the seed lists of k_seed_scan equal
    the scan of those bits"""
    import subprocess
    import desamba_amd as D
    idx = D.Index(demo["index"]); ctx = D.Ctx(idx, 0)
    try:
        fill = 0.2; table_bytes = 1 << 31; k = 18; mask = (1 << 34) - 1; sbm = int(0.8 * k)
        ctx.use_synthetic_filter(table_bytes, fill)
        fq = tmp_path / "syn.fq"
        subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "96", "3000", "0.15", "77", "ont"])
        recs = D.read_fastq(str(fq)) + [(b"polyA", b"A" * 200, None), (b"short", b"ACGT" * 9, None)]
        reads = D.make_reads(recs)
        got = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("DSB_SEED_SCAN", mode)
            ctx.reset_history(); ctx.reload_env(); ctx.upload(reads); ctx.run()
            t = ctx.timing()
            assert t.seed_scan == int(mode) and t.classify_ms == 0
            got[mode] = [(ctx.seeds(i, 1), ctx.seeds(i, 0)) for i in range(len(recs))]
        assert got["0"] == got["1"]
        assert sum(len(a[0]) + len(b[0]) for a, b in got["1"]) > 1000
        L = D.lib()
        code = {65: 0, 97: 0, 71: 2, 103: 2, 84: 3, 116: 3}
        for i in (0, 17, 96):
            seq = recs[i][1]; n = len(seq) - k + 1
            b = [code.get(c, 1) for c in seq]
            for strand in (1, 0):
                st = b if strand else [3 - x for x in reversed(b)]
                exp = bytearray(n)
                for w in range(n):
                    km = 0; cnt = [0, 0, 0, 0]
                    for x in st[w:w + k]:
                        km = (km << 2) | x; cnt[x] += 1
                    if km == 0 or max(cnt) >= sbm:
                        continue
                    h1 = _hash64_1(km) & mask
                    if not L.dsb_synthetic_filter_bit(0, h1, fill):
                        continue
                    exp[w] = 1 if L.dsb_synthetic_filter_bit(1, _hash64_2(km) & mask, fill) else 0
                assert ctx.exist_bits(i, strand) == bytes(exp), (i, strand)
    finally:
        ctx.close(); idx.close()


@pytest.mark.parametrize("name", ["heavy", "wrapq", "ont20k", "ont5k_e25", "manyanchors"])
def test_heavy_reads_on_several_wavefronts(gpu, name, monkeypatch):
    """k_classify_heavy: the very heaviest reads of a batch run on a workgroup of eight wavefronts each (wave 0 runs the read,
    the others split the old-predecessor pass of the batched sparse DP); forced onto small golden sets"""
    D, idx, ctx = gpu
    monkeypatch.setenv("DSB_HEAVY_FIRST", "16"); monkeypatch.setenv("DSB_HEAVY_MW", "16")
    recs = D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq"))
    exp = open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()
    if len(recs) < 4:           # (the early launch takes at most half of a batch: feed the set three times)
        recs = recs * 3; exp = exp * 3
    hits, sam = classify_all(D, ctx, recs)
    assert ctx.timing().n_heavy_mw > 0
    assert sam == exp


def test_heavy_reads_on_several_wavefronts_second_index(strain, monkeypatch):
    """the same on the strain index, whose reads have match-node lists of tens of thousands of nodes (the multi-wave DP pass
    does run there)"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_HEAVY_FIRST", "48"); monkeypatch.setenv("DSB_HEAVY_MW", "48")
    idx = D.Index(strain["index"]); ctx = D.Ctx(idx, 0)
    try:
        hits, sam = classify_all(D, ctx, D.read_fastq(strain["fastq"]))
        assert ctx.timing().n_heavy_mw == 48
        assert sam == open(strain["sam"], "rb").read()
    finally:
        ctx.close(); idx.close()


@pytest.mark.parametrize("scan", ["0", "1"])
def test_heavy_reads_are_handed_over(strain, monkeypatch, scan):
    """DSB_ST_HEAVY: a read whose sparse DP scans more than heavy_limit predecessors on its one wavefront is given up there,
    listed on the device and run again by a workgroup of eight wavefronts (k_classify_heavy<8>) after the main launch --
    with the limit forced down (DSB_HEAVY_PREDS) so that most reads of the strain set go that way, from hit bits
    (DSB_SEED_SCAN=0) and from the seed lists of k_seed_scan: same SAM, nothing left with a status"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_HEAVY_PREDS", "20000"); monkeypatch.setenv("DSB_SEED_SCAN", scan)
    monkeypatch.setenv("DSB_HEAVY_FIRST", "8"); monkeypatch.setenv("DSB_HEAVY_MW", "4")
    idx = D.Index(strain["index"]); ctx = D.Ctx(idx, 0)
    try:
        recs = D.read_fastq(strain["fastq"])
        hits, sam = classify_all(D, ctx, recs)
        t = ctx.timing()
        assert t.n_requeue > len(recs) // 4            # (a few of them outgrow the match-node arena there and take the second run as well)
        assert sam == open(strain["sam"], "rb").read()
        monkeypatch.setenv("DSB_HEAVY_PREDS", "0")                      # switched off: nothing is handed over
        hits, sam = classify_all(D, ctx, recs)
        assert ctx.timing().n_requeue == 0 and sam == open(strain["sam"], "rb").read()
    finally:
        ctx.close(); idx.close()


@pytest.mark.parametrize("name", ["heavy", "manyanchors", "ont20k"])
def test_heavy_hand_over_on_the_demo_index(gpu, name, monkeypatch):
    """the same on golden sets of the demo index (hand-over from the right/left extension loops and from a large middle gap)"""
    D, idx, ctx = gpu
    monkeypatch.setenv("DSB_HEAVY_PREDS", "5000")
    recs = D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq"))
    exp = open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()
    hits, sam = classify_all(D, ctx, recs)
    assert ctx.timing().n_requeue > 0
    assert sam == exp


def test_short_reads_64_per_wavefront(gpu, demo, oracle, tmp_path, monkeypatch):
    """group mode of k_classify (batches of reads <= 400 bases with seed lists from k_seed_scan): the anchor stage of one read
    per lane, 64 reads per work item.  6000 x 150 bp at 1 % and 2000 at 8 % error (slow path, reads without hits), plus
    reads shorter than 40: SAM byte-identical to the oracle's"""
    import subprocess
    D, idx, ctx = gpu
    a = tmp_path / "a.fq"; b = tmp_path / "b.fq"; c = tmp_path / "c.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(a), "6000", "150", "0.01", "901", "ngs"])
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(b), "2000", "150", "0.08", "902", "ngs"])
    c.write_bytes(a.read_bytes() + b"@s1\nACGTACGTACGTACGTACGTACGT\n+\n555555555555555555555555\n@s2\n" + b"A" * 300 + b"\n+\n" + b"5" * 300 + b"\n" + b.read_bytes())
    recs = D.read_fastq(str(c))
    monkeypatch.delenv("DSB_SEED_SCAN", raising=False)
    hits, sam = classify_all(D, ctx, recs)
    assert ctx.timing().seed_scan == 1
    exp = tmp_path / "exp.sam"
    oracle.classify_file(str(c), str(exp), threads=4)
    assert sam == exp.read_bytes()
    monkeypatch.setenv("DSB_NO_GROUP", "1")
    hits2, sam2 = classify_all(D, ctx, recs)
    assert sam2 == sam


@pytest.mark.gpu
def test_hinted_ctx_allocates_nothing_per_batch(demo, capfd, monkeypatch):
    """dsb_ctx_create with hints sizes everything; no batch may allocate afterwards.  (Round 3: the four-read warm-up batch
    of a ctx hinted for 100-kbase reads counted its arenas as oversized and gave them back -- the first real batch rebuilt
    them on the per-batch path, which stalls a sibling context for seconds.)"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_UPLOAD_TRACE", "1")
    idx = D.Index(demo["index"])
    ctx = D.Ctx(idx, 0, max_read_len=100064, max_batch_reads=8192, max_batch_bases=8192 * 12000)
    capfd.readouterr()                                   # (what the set-up allocated)
    recs = D.read_fastq(os.path.join(GOLDEN, "synth", "pb.fq")) + D.read_fastq(demo["fastq"], 200)      # mixed lengths, all far below the hint
    for part in (recs[: len(recs) // 2], recs[len(recs) // 2:], recs):
        res = ctx.classify(D.make_reads(part))
        assert all(res.reads[i].status == 0 for i in range(len(part)))
    err = capfd.readouterr().err
    assert "[upload]" in err                             # the trace is on
    assert "grew" not in err and "(re)built" not in err, err
    ctx.close(); idx.close()


@pytest.mark.gpu
def test_cli_reads_from_a_fifo(gpu, demo, golden_md5, tmp_path):
    """A FIFO as input (the reference reads whatever gzopen opens).  Round 3: the look at the input before the contexts are
    made opened the FIFO and closed it again -- the writer died of SIGPIPE and the run waited for it for ever."""
    import subprocess
    import threading
    fifo = tmp_path / "reads.fifo"
    os.mkfifo(fifo)
    out = tmp_path / "fifo.sam"
    data = open(demo["fastq"], "rb").read()

    def feed():
        with open(fifo, "wb") as f:
            f.write(data)
    t = threading.Thread(target=feed, daemon=True)
    t.start()
    cli = os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA")
    p = subprocess.run([cli, "classify", demo["index"], str(fifo), "-o", str(out)], stderr=subprocess.PIPE, timeout=120)
    t.join(10)
    assert p.returncode == 0, p.stderr
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == golden_md5
