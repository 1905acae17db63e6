"""Index construction (`deSAMBA index`): the files written must be those of the reference's builder.

Golden digests (tests/golden/build/*.md5.json, made by tests/golden/make_build_golden.py from the reference binary)
for small synthetic references, and the demo index's md5s.  CPU tests run the builder's stages in the
host emulation (tests/emu/emu_build.cpp: the same dsb_build_impl.h the GPU runs); GPU tests go through
dsb_index_build / the CLI."""
import gzip
import hashlib
import json
import os
import subprocess
import zipfile

import pytest

import build_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "build")
CASES = ["graph1", "graph2", "graph3", "reader"]
# md5 of the demo index files as the reference builds them (stable across rebuilds)
DEMO_MD5 = {".bwt": "1532b15ccccbf7268386e2758000f5c2", ".exk0": "000472fabf9c519d075726634427736f", ".exk1": "bbbd736229d6b4a3e21c494f094c0ac4",
            ".exki": "02ae172ff75582d3f4ffda7001e7d849", ".sa": "aaef536051f4d95e1370844a1be230dc", ".ref_b": "8e603575e9d0e4116e7d2b3e3369a750",
            ".ref_p": "00ef48f10125466085d1d8149579e0e1", ".unv": "14f1f3e67a4460e8c925c7108c740533", ".acg": "0c231610f8892d7042b5ea5dd4eef4ec"}


def md5_file(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def demo_fasta(tmp_path):
    zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", "demo", "viral-gs.zip")).extractall(tmp_path)
    return str(tmp_path / "viral-gs.fa")


def check_case(name, out):
    want = json.load(open(os.path.join(GOLD, name + ".md5.json")))
    got = build_lib.digest_dir(out)
    assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}


@pytest.mark.parametrize("name", CASES)
def test_emulated_stages_write_the_reference_files(name, tmp_path):
    """branching graphs, reverse complements, N runs, repeats, runs of exactly 31 / 30 bases, empty and tiny records,
    duplicates (graph*); CRLF, blank lines, '@' headers, a FASTQ record (reader): gzip FASTA in, every file's digest out"""
    out = str(tmp_path / "idx")
    build_lib.emu_build(os.path.join(GOLD, name + ".fa.gz"), out)
    check_case(name, out)


def test_supplied_kmer_list_gives_the_same_index(tmp_path):
    """`index kmer.srt ref.fa dir` (the reference's form): the k-mer list is read, not enumerated"""
    text = gzip.open(os.path.join(GOLD, "graph1.fa.gz")).read()
    srt = str(tmp_path / "kmer.srt")
    build_lib.write_kmer_srt_from_text(build_lib.reader_view(text), srt)
    out = str(tmp_path / "idx")
    build_lib.emu_build(os.path.join(GOLD, "graph1.fa.gz"), out, kmer_srt=srt)
    check_case("graph1", out)


def test_kmer_missing_from_a_supplied_list_is_an_error(tmp_path):
    """the reference asserts (binSearch, src/idx.c:84-96); here the build returns an error"""
    text = gzip.open(os.path.join(GOLD, "graph1.fa.gz")).read()
    recs = build_lib.reader_view(text)
    srt = str(tmp_path / "kmer.srt")
    build_lib.write_kmer_srt_from_text(recs[1:], srt)
    with pytest.raises(RuntimeError):
        build_lib.emu_build(os.path.join(GOLD, "graph1.fa.gz"), str(tmp_path / "idx"), kmer_srt=srt)


@pytest.mark.parametrize("text", [b">short\nACGTACGTACGTACGTACGT\n", b">no_kmers\n" + b"ACGTACGTAC" * 2 + b"N" + b"TTGCA" * 5 + b"\n>n\n" + b"N" * 100 + b"\n"])
def test_references_without_a_31_mer_are_refused(text, tmp_path):
    """a reference shorter than 31 bases, or without any ACGT run of 31: an error code, not a crash (the reference's builder
    reads an empty kmer.srt and asserts later)"""
    fa = tmp_path / "r.fa"; fa.write_bytes(text)
    with pytest.raises(RuntimeError):
        build_lib.emu_build(str(fa), str(tmp_path / "idx"))


def test_reader_view_of_the_awkward_fasta():
    """the plain-Python restatement of the reader used to make the reader case's k-mer list sees what the reference saw
    (names and lengths are pinned by the golden .ref_i digest; this spells them out)"""
    recs = build_lib.reader_view(gzip.open(os.path.join(GOLD, "reader.fa.gz")).read())
    names = [n for n, _ in recs[:5]]
    assert names == [b"crlf", b"blank", b"atsign", b"fastq", b"after"]
    assert [len(s) for _, s in recs[:5]] == [70 + 1 + 64 + 1, 80 + 1 + 17 + 75, 140, 120, 133]       # '\r' kept; '\n' + the swallowed header line


def test_emulated_stages_rebuild_the_demo_index(tmp_path):
    """the 463-genome demo reference (11.5 Mbp): all nine md5s of SURVEY.md 8c"""
    out = str(tmp_path / "idx")
    st = build_lib.emu_build(demo_fasta(tmp_path), out)
    assert st[0] == 10982489 and st[3] == 463
    got = {e: md5_file(os.path.join(out, "deSAMBA" + e)) for e in DEMO_MD5}
    assert got == DEMO_MD5


# ---- the builder in ranges of k-mer prefixes (dsb_build_parts.h): what a build beyond the device's memory runs ---------------------

@pytest.mark.parametrize("name,parts", [("graph2", 5), ("graph3", 7)])       # (graph1 in ranges: the supplied-list test below; all four cases in 3 ranges: the GPU tests)
def test_build_in_ranges_writes_the_reference_files(name, parts, tmp_path):
    """dsb_build_run_parts on the host: the k-mer, unitig-number and BWT-row stages each in `parts` ranges of 13-mer prefixes (the
    golden references have unitigs that start in one range and end in another, k-mers whose neighbours lie in other ranges, padded
    suffixes that sort between the k-mers of another range): every file's digest as the reference's builder wrote it"""
    out = str(tmp_path / "idx")
    st = build_lib.emu_build_parts(os.path.join(GOLD, name + ".fa.gz"), out, parts=parts)
    assert st["parts_kmers"] >= parts and st["parts_rows"] >= parts and st["parts_uid"] >= min(parts, 3)
    assert st["n_rows"] == json.load(open(os.path.join(GOLD, name + ".md5.json")))["n_rows"]
    check_case(name, out)


def test_build_in_ranges_with_the_kmer_list_in_a_file(tmp_path, monkeypatch):
    """DSB_BUILD_SPILL=1: the k-mer list of the ranges (8 bytes per 31-mer: what a 35-Gbp build would otherwise hold in host memory) is appended
    to <IndexDir>/deSAMBA.kmers.tmp range by range, mapped for the later stages and removed at the end: the same files"""
    monkeypatch.setenv("DSB_BUILD_SPILL", "1")
    out = str(tmp_path / "idx")
    st = build_lib.emu_build_parts(os.path.join(GOLD, "graph3.fa.gz"), out, parts=5)
    assert st["spilled_bytes"] == 8 * st["n_kmer"] and not os.path.exists(os.path.join(out, "deSAMBA.kmers.tmp"))
    check_case("graph3", out)


def test_build_in_ranges_with_a_supplied_kmer_list(tmp_path):
    """`index kmer.srt ref.fa dir` in ranges: the list is cut at the prefixes, a k-mer of the text that the list lacks is an error"""
    text = gzip.open(os.path.join(GOLD, "graph1.fa.gz")).read()
    recs = build_lib.reader_view(text)
    srt = str(tmp_path / "kmer.srt")
    build_lib.write_kmer_srt_from_text(recs, srt)
    out = str(tmp_path / "idx")
    build_lib.emu_build_parts(os.path.join(GOLD, "graph1.fa.gz"), out, kmer_srt=srt, parts=4)
    check_case("graph1", out)
    build_lib.write_kmer_srt_from_text(recs[1:], srt)
    with pytest.raises(RuntimeError):
        build_lib.emu_build_parts(os.path.join(GOLD, "graph1.fa.gz"), str(tmp_path / "idx2"), kmer_srt=srt, parts=4)


def test_build_in_ranges_under_a_budget_rebuilds_the_demo_index(tmp_path):
    """the demo reference (11.5 Mbp, 11 M 31-mers) under a budget of 400 MB -- a fifth of what the one-piece build holds on the host
    backend (whose fixed tables of 2^26 entries alone are 1.3 GB): the ranges are planned from the budget, the backend counts every
    allocation, the nine md5s of SURVEY.md 8c come out"""
    out = str(tmp_path / "idx")
    budget = 400 << 20
    st = build_lib.emu_build_parts(demo_fasta(tmp_path), out, budget=budget)
    assert st["n_kmer"] == 10982489 and st["n_refs"] == 463
    assert st["peak"] <= budget, st
    assert st["parts_kmers"] >= 2 and st["parts_rows"] >= 2, st
    got = {e: md5_file(os.path.join(out, "deSAMBA" + e)) for e in DEMO_MD5}
    assert got == DEMO_MD5
    with pytest.raises(RuntimeError):                       # a budget that does not hold the text + slack: an error code, nothing built
        build_lib.emu_build_parts(demo_fasta(tmp_path), str(tmp_path / "idx2"), budget=40 << 20)


# ---------------------------------------------------------------- GPU: through the C ABI and the CLI

@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_build_writes_the_reference_files(name, tmp_path):
    import desamba_amd as D
    out = str(tmp_path / "idx")
    st = D.build_index(os.path.join(GOLD, name + ".fa.gz"), out)
    assert st.n_rows == json.load(open(os.path.join(GOLD, name + ".md5.json")))["n_rows"]
    check_case(name, out)


@pytest.mark.gpu
def test_gpu_build_of_the_demo_index_and_classify_on_it(tmp_path):
    """dsb_index_build on the demo reference: the nine md5s; then `deSAMBA classify` of the demo reads on THAT index
    gives the demo SAM (md5 1da908b6...)"""
    import desamba_amd as D
    fa = demo_fasta(tmp_path)
    out = str(tmp_path / "idx")
    st = D.build_index(fa, out)
    print("demo index built in %.2f s (read %.2f, k-mers %.2f, graph %.2f, unitigs %.2f, rows %.2f, tables %.2f, write %.2f)" %
          (st.total_s, st.parse_s, st.sort_s, st.graph_s, st.walk_s, st.rows_s, st.tables_s, st.write_s))
    assert st.n_kmer == 10982489 and st.n_refs == 463
    got = {e: md5_file(os.path.join(out, "deSAMBA" + e)) for e in DEMO_MD5}
    assert got == DEMO_MD5
    zipfile.ZipFile(os.path.join(ROOT, "tests", "golden", "demo", "ERR1050068.zip")).extractall(tmp_path)
    sam = str(tmp_path / "out.sam")
    subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify", out, str(tmp_path / "ERR1050068.fastq"), "-o", sam], check=True, stderr=subprocess.DEVNULL)
    assert md5_file(sam) == open(os.path.join(ROOT, "tests", "golden", "demo_sam.md5")).read().split()[0]


@pytest.mark.gpu
def test_gpu_build_errors_are_codes_not_crashes(tmp_path):
    """a k-mer of the text missing from the supplied list, a reference without any 31-mer, a missing file: error codes (the
    device memory of a failed build is released with its backend); a good build still works afterwards"""
    import desamba_amd as D
    text = gzip.open(os.path.join(GOLD, "graph1.fa.gz")).read()
    srt = str(tmp_path / "kmer.srt")
    build_lib.write_kmer_srt_from_text(build_lib.reader_view(text)[1:], srt)
    with pytest.raises(D.DsbError):
        D.build_index(os.path.join(GOLD, "graph1.fa.gz"), str(tmp_path / "a"), kmer_srt=srt)
    (tmp_path / "short.fa").write_bytes(b">s\nACGTACGTAC\n")
    with pytest.raises(D.DsbError):
        D.build_index(str(tmp_path / "short.fa"), str(tmp_path / "b"))
    with pytest.raises(D.DsbError):
        D.build_index(str(tmp_path / "missing.fa"), str(tmp_path / "c"))
    D.build_index(os.path.join(GOLD, "graph1.fa.gz"), str(tmp_path / "d"))
    check_case("graph1", str(tmp_path / "d"))


@pytest.mark.gpu
def test_cli_index_with_and_without_a_kmer_list(tmp_path):
    """`deSAMBA index [SortedKmer] <Reference> <IndexDir>` (build_index_main, src/idx.c:1254-1282)"""
    cli = os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA")
    fa = os.path.join(GOLD, "graph2.fa.gz")
    subprocess.run([cli, "index", fa, str(tmp_path / "a")], check=True, stderr=subprocess.DEVNULL)
    check_case("graph2", str(tmp_path / "a"))
    srt = str(tmp_path / "kmer.srt")
    build_lib.write_kmer_srt_from_text(build_lib.reader_view(gzip.open(fa).read()), srt)
    subprocess.run([cli, "index", srt, fa, str(tmp_path / "b")], check=True, stderr=subprocess.DEVNULL)
    check_case("graph2", str(tmp_path / "b"))
    assert subprocess.run([cli, "index", str(tmp_path / "nope.fa"), str(tmp_path / "c")], stderr=subprocess.DEVNULL).returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_build_in_ranges_writes_the_reference_files(name, tmp_path, monkeypatch):
    """dsb_index_build with DSB_BUILD_PARTS=3: the three per-range stages of dsb_build_parts.h as HIP kernels (atomic appends, rocPRIM
    sorts of one range at a time, text passes with the start list): the reference's files"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_BUILD_PARTS", "3")
    out = str(tmp_path / "idx")
    st = D.build_index(os.path.join(GOLD, name + ".fa.gz"), out)
    assert st.ranges_kmers >= 3 and st.ranges_rows >= 3 and st.budget_bytes > 0
    check_case(name, out)


@pytest.mark.gpu
def test_gpu_build_budget_too_small_is_an_error_code(tmp_path, monkeypatch):
    """a budget that does not even hold the text and the fixed tables: DSB_ENOMEM (-3), nothing written, and the next build works"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_BUILD_BUDGET", "16m")
    with pytest.raises(D.DsbError) as e:
        D.build_index(os.path.join(GOLD, "graph3.fa.gz"), str(tmp_path / "a"))
    assert e.value.code == -3 and not os.path.exists(str(tmp_path / "a" / "deSAMBA.bwt"))
    monkeypatch.setenv("DSB_BUILD_BUDGET", "2g")
    st = D.build_index(os.path.join(GOLD, "graph3.fa.gz"), str(tmp_path / "b"))
    assert st.budget_bytes == 2 << 30 and st.peak_device_bytes <= st.budget_bytes
    check_case("graph3", str(tmp_path / "b"))


@pytest.mark.gpu
def test_gpu_build_beyond_the_budget_gives_the_same_files(tmp_path, monkeypatch):
    """a 380-Mbp collection (tandem repeats, strains, mobile elements: 4e8 BWT rows) built in one piece and under
    DSB_BUILD_BUDGET = an eighth of what the one-piece build held: the same ten files byte for byte, and the device never held more
    than the budget (every allocation of the build is counted)"""
    import shutil
    import desamba_amd as D
    if shutil.disk_usage(str(tmp_path)).free < (8 << 30):
        pytest.skip("needs 8 GiB of disk")
    fa = str(tmp_path / "syn.fa"); a = str(tmp_path / "a"); b = str(tmp_path / "b")
    subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, "380", "1"], check=True, stderr=subprocess.DEVNULL)
    s1 = D.build_index(fa, a)
    assert s1.budget_bytes == 0 and s1.peak_device_bytes > 30 * s1.n_bases
    budget = s1.peak_device_bytes // 8
    monkeypatch.setenv("DSB_BUILD_BUDGET", str(budget))
    monkeypatch.setenv("DSB_BUILD_SPILL", "1")              # ... and the k-mer list through a file instead of host memory
    s2 = D.build_index(fa, b)
    monkeypatch.delenv("DSB_BUILD_BUDGET"); monkeypatch.delenv("DSB_BUILD_SPILL")
    assert s2.spilled_bytes == 8 * s2.n_kmer and not os.path.exists(os.path.join(b, "deSAMBA.kmers.tmp"))
    print("380 Mbp: one piece %.1f s holding %.2f GiB; budget %.2f GiB: %.1f s holding %.2f GiB in %d / %d / %d / %d passes (k-mers / unitig numbers / rows / filter tables)" %
          (s1.total_s, s1.peak_device_bytes / 2**30, budget / 2**30, s2.total_s, s2.peak_device_bytes / 2**30, s2.ranges_kmers, s2.ranges_unitig_numbers, s2.ranges_rows, s2.ranges_exist))
    assert s2.budget_bytes == budget and s2.peak_device_bytes <= budget and s2.ranges_kmers >= 4
    assert (s2.n_kmer, s2.n_unitig, s2.n_rows) == (s1.n_kmer, s1.n_unitig, s1.n_rows)
    for e in build_lib.EXTS:
        assert md5_file(os.path.join(a, "deSAMBA" + e)) == md5_file(os.path.join(b, "deSAMBA" + e)), e
    shutil.rmtree(a); shutil.rmtree(b)


# ---- every filter k the reference knows (src/idx.c:966-982): forced table levels on one small reference -----------------------------
# tests/golden/eklevel (made by make_eklevel_golden.py from the reference's UB-pinned build with DSB_FORCE_EK_LEVEL): level 1 = k 17,
# 2 x 256 MiB, 31-bit mask; level 5 = k 19, 2 x 4 GiB, 35-bit mask; level 7 = k 20, 2 x 16 GiB, 37-bit mask (rounds 1-3 only ever met k 16 / 18)
EKL = os.path.join(ROOT, "tests", "golden", "eklevel")


def eklevel_inputs(tmp_path):
    fa = str(tmp_path / "ref.fa"); fq = str(tmp_path / "reads.fq")
    open(fa, "wb").write(gzip.open(os.path.join(EKL, "ref.fa.gz")).read())
    open(fq, "wb").write(gzip.open(os.path.join(EKL, "reads.fq.gz")).read())
    return fa, fq


def check_eklevel_files(out, lv, hash_tables=True):
    want = json.load(open(os.path.join(EKL, "digests.json")))[str(lv)]
    assert os.path.getsize(os.path.join(out, "deSAMBA.exk0")) == want["exk_bytes"] == os.path.getsize(os.path.join(out, "deSAMBA.exk1"))
    got = {e: hashlib.md5(build_lib.canonical_bytes(out, e)).hexdigest() for e in build_lib.EXTS if e not in (".exk0", ".exk1")}
    if hash_tables:
        got[".exk0"] = md5_file(os.path.join(out, "deSAMBA.exk0")); got[".exk1"] = md5_file(os.path.join(out, "deSAMBA.exk1"))
    assert got == {k: want[k] for k in got}, {k: (got[k], want[k]) for k in got if got[k] != want[k]}


def test_forced_k17_tables_in_the_host_emulation_and_the_oracle(tmp_path, monkeypatch):
    """level 1 (k = 17): the builder's stages on the host write the reference's files; the oracle classifies on them like the reference"""
    import oracle_lib
    fa, fq = eklevel_inputs(tmp_path)
    monkeypatch.setenv("DSB_FORCE_EK_LEVEL", "1")
    out = str(tmp_path / "idx")
    build_lib.emu_build(fa, out)
    check_eklevel_files(out, 1)
    sam = str(tmp_path / "o.sam")
    oracle_lib.Oracle(out).classify_file(fq, sam)
    assert open(sam, "rb").read() == open(os.path.join(EKL, "level1.ubfree.sam"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("lv", [1, 5, 7])
def test_gpu_forced_filter_levels(lv, tmp_path, monkeypatch):
    """dsb_index_build with k = 17 / 19 / 20 filter tables forced on a small reference: the reference's files (the 16-GiB tables of
    level 7 by size only: their content is checked through what is classified on them), and `deSAMBA classify` on that index gives the
    SAM of the reference on ITS index of that level -- 36- and 37-bit hash masks, 64-bit table offsets, k-mers of up to 40 bits"""
    import shutil
    import desamba_amd as D
    need = 2 * (1 << (27 + lv))
    free_disk = shutil.disk_usage(str(tmp_path)).free
    avail = next(int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable:"))
    if free_disk < need * 1.2 + (2 << 30) or avail < need * 1.5 + (4 << 30):
        pytest.skip("level %d needs %.0f GiB of disk and memory: %.0f / %.0f GiB free" % (lv, need / 2**30, free_disk / 2**30, avail / 2**30))
    fa, fq = eklevel_inputs(tmp_path)
    monkeypatch.setenv("DSB_FORCE_EK_LEVEL", str(lv))
    out = str(tmp_path / "idx")
    D.build_index(fa, out)
    monkeypatch.delenv("DSB_FORCE_EK_LEVEL")
    check_eklevel_files(out, lv, hash_tables=(lv < 7))
    sam = str(tmp_path / "out.sam")
    subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify", out, fq, "-o", sam], check=True, stderr=subprocess.DEVNULL)
    assert open(sam, "rb").read() == open(os.path.join(EKL, "level%d.ubfree.sam" % lv), "rb").read()
    shutil.rmtree(out)


@pytest.mark.gpu
def test_gpu_gbp_scale_index_against_the_reference(tmp_path):
    """BASELINE configs[4] shape under the test runner: a 1.2-Gbp synthetic collection indexed by dsb_index_build (1 G BWT rows, filter
    tables of 2 x 512 MiB with k = 17, the raw or compressed 13-mer table as the data decide), 2048 PacBio-mixed reads classified by
    the CLI and by the reference's UB-pinned build on the same index directory: identical SAM"""
    import shutil
    import desamba_amd as D
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "deSAMBA_ubfree")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/deSAMBA_ubfree is not built")
    if shutil.disk_usage(str(tmp_path)).free < (12 << 30):
        pytest.skip("needs 12 GiB of disk")
    fa = str(tmp_path / "syn.fa"); out = str(tmp_path / "idx"); fq = str(tmp_path / "r.fq")
    subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, "1200", "17", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
    st = D.build_index(fa, out)
    os.remove(fa)
    print("1.2-Gbp index: %d sequences, %d 31-mers, %d BWT rows in %.1f s" % (st.n_refs, st.n_kmer, st.n_rows, st.total_s))
    assert st.n_bases > 1_150_000_000 and os.path.getsize(os.path.join(out, "deSAMBA.exk0")) >= (256 << 20)
    subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "gen_fastq.py"), out, fq, "2048", "12000", "0.13", "3", "pacbio", "8"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify", out, fq, "-o", str(tmp_path / "gpu.sam")], check=True, stderr=subprocess.DEVNULL)
    t = min(16, len(os.sched_getaffinity(0)))
    subprocess.run([ref_bin, "classify", "-t", str(t), out, fq, "-o", str(tmp_path / "ref.sam")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    a = open(tmp_path / "gpu.sam", "rb").read(); b = open(tmp_path / "ref.sam", "rb").read()
    assert a.count(b"\n") > 2048 and a == b
    shutil.rmtree(out)


@pytest.mark.gpu
def test_gpu_headline_index_reads_decided_by_the_read_buffer_pad(tmp_path):
    """tests/golden/synth/u1_flagpair.*: the two reads of the headline batch whose printed secondaries differ between the stock reference
    and its UB-pinned build (one pad byte shifts every score of the read by 1, the reference's parity tie-break then prints another
    five of a dozen tied hits: u1_flagpair.txt).  On the headline index, built here, this repo prints the UB-pinned lines."""
    import desamba_amd as D
    fa = str(tmp_path / "syn.fa"); out = str(tmp_path / "idx")
    subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, "320", "11", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
    D.build_index(fa, out)
    os.remove(fa)
    g = os.path.join(ROOT, "tests", "golden", "synth", "u1_flagpair")
    sam = str(tmp_path / "o.sam")
    subprocess.run([os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA"), "classify", out, g + ".fq", "-o", sam], check=True, stderr=subprocess.DEVNULL)
    got = open(sam, "rb").read()
    assert got == open(g + ".ubfree.sam", "rb").read()
    assert got != open(g + ".stock_t1.sam", "rb").read()
