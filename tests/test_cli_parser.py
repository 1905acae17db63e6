"""The CLI's input side (desamba_amd/csrc/desamba_main.c: pinned-buffer filling, carry-over between buffers, in-place
record parser) against an independent character-level restatement of the reference's kseq_read (src/lib/utils.c:939-977,
the old kseq that keeps '\\r' and takes the first character of every sequence line as data), on awkward inputs and with
buffers barely larger than one record; tests/test_oracle_golden.py::test_kseq_rules_against_reference pins the
restatement to the compiled reference.  No GPU."""
import gzip
import os
import random
import re
import subprocess

import pytest

from conftest import ROOT


def kseq_records(data: bytes):
    """character-level restatement of the reference's kseq_read (the OLD kseq: ks_getc / ks_getuntil2 with '\\n' as the
    only line delimiter) driven as read_reads drives it (src/cly_mt.c:42-56: a record whose kseq_read returns -2 is
    dropped and reading goes on behind it): yields (name, seq, qual or None)"""
    pos = 0; n = len(data); last = 0

    def getc():
        nonlocal pos
        if pos >= n:
            return -1
        c = data[pos]; pos += 1
        return c

    def getuntil(delims, out):
        """ks_getuntil2: append up to the first delimiter (consumed, not appended); -1 only at EOF with nothing read;
        returns (code, delimiter or 0)"""
        nonlocal pos
        if pos >= n:
            return -1, 0
        while pos < n:
            c = data[pos]; pos += 1
            if c in delims:
                return 0, c
            out.append(c)
        return 0, 0

    while True:
        if last == 0:
            c = getc()
            while c != -1 and c not in (62, 64):
                c = getc()
            if c == -1:
                return
            last = c
        name = bytearray(); seq = bytearray(); qual = bytearray()
        rc, c = getuntil(b" \t\n\r\x0b\x0c", name)
        if rc < 0:
            return
        if c != 10:
            getuntil(b"\n", bytearray())
        c = getc()
        while c != -1 and c not in (62, 43, 64):
            seq.append(c)                      # whatever it is, '\n' and '\r' included
            getuntil(b"\n", seq)
            c = getc()
        if c in (62, 64):
            last = c
        if c != 43:
            yield bytes(name), bytes(seq), None
            continue
        c = getc()
        while c != -1 and c != 10:
            c = getc()
        if c == -1:
            return                             # -2 at the end of the input
        while True:
            rc, _ = getuntil(b"\n", qual)
            if rc < 0 or len(qual) >= len(seq):
                break
        last = 0
        if len(qual) != len(seq):
            continue                           # -2: the record is dropped
        yield bytes(name), bytes(seq), bytes(qual)


@pytest.fixture(scope="module")
def harness(built, tmp_path_factory):
    exe = tmp_path_factory.mktemp("cli") / "parse_harness"
    # DSB_HARNESS_CFLAGS="-fsanitize=address,undefined -g": the reader under the sanitizers (tests/tools/cli_sanitize.sh)
    subprocess.check_call(["gcc", "-O2", "-std=gnu99"] + os.environ.get("DSB_HARNESS_CFLAGS", "").split() + ["-I" + os.path.join(ROOT, "include"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "cli", "parse_harness.c"), "-L" + os.path.join(ROOT, "desamba_amd"), "-ldesamba_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "desamba_amd"), "-lpthread", "-lz", "-ldl"])
    return str(exe)


def run_harness(harness, cap, path):
    out = subprocess.run([harness, str(cap), path], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    def unesc(b):
        return b.replace(b"\\n", b"\n").replace(b"\\r", b"\r").replace(b"\\t", b"\t").replace(b"\\\\", b"\\") if b"\\" in b else b
    recs = []
    for line in out.split(b"\n")[:-1]:
        name, seq, qual, hist = line.split(b"\t")
        recs.append((unesc(name), unesc(seq), unesc(qual), int(hist)))
    return recs


def make_inputs():
    rnd = random.Random(7)
    def dna(k):
        return bytes(rnd.choice(b"ACGTN") for _ in range(k))
    def q(k):   # '@', '>' and '+' are legal quality characters, also at the start of a line
        return bytes(rnd.choice(b"@>+5I#!") for _ in range(k))
    four = b"".join(b"@r%d some text\n%s\n+\n%s\n" % (i, s, q(len(s))) for i, s in enumerate(dna(rnd.randint(1, 300)) for _ in range(60)))
    crlf = four.replace(b"\n", b"\r\n")
    multi_fa = b"".join(b">c%d\n" % i + b"\n".join(dna(rnd.randint(1, 60)) for _ in range(rnd.randint(1, 5))) + b"\n" for i in range(40))
    multi_fq = b""
    for i in range(40):
        lines = [dna(rnd.randint(1, 50)) for _ in range(rnd.randint(1, 4))]; tot = sum(map(len, lines)); qq = q(tot); cut = sorted(rnd.sample(range(1, tot), min(2, tot - 1))) if tot > 2 else []
        qs = [qq[a:b] for a, b in zip([0] + cut, cut + [tot])]
        multi_fq += b"@m%d\n" % i + b"\n".join(lines) + b"\n+m%d\n" % i + b"\n".join(qs) + b"\n"
    junk = b"junk before\n\n" + four[:2000].rsplit(b"\n@", 1)[0] + b"\n\n\n" + multi_fa
    no_nl = four.rstrip(b"\n")
    empty_seq = b"@e1\n\n+\n\n@e2\nACGT\n+\n5555\n>f1\n>f2\nAC\n"
    return {"four": four, "crlf": crlf, "multi_fa": multi_fa, "multi_fq": multi_fq, "junk": junk, "no_nl": no_nl, "empty_seq": empty_seq, "mixed": multi_fa + four + multi_fq}


@pytest.mark.parametrize("name", sorted(make_inputs()))
def test_parser_matches_kseq(harness, tmp_path, name):
    data = make_inputs()[name]
    exp = list(kseq_records(data))
    assert exp, name
    longest = max(len(n) + 2 * len(s) for n, s, _ in exp) + 700
    path = tmp_path / (name + ".fq"); path.write_bytes(data)
    gz = tmp_path / (name + ".fq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(data)
    for cap in (longest, longest + 13, 4096, 1 << 20):
        for p in (path, gz):
            got = run_harness(harness, cap, str(p))
            assert [(n, s, q if q is not None else b"") for n, s, q in exp] == [(n, s, q) for n, s, q, _ in got], (name, cap, str(p))
            hist = 0
            # hist_before is constant within a batch: the running maximum at the first record of the batch
            seen = []
            for (n, s, q, h) in got:
                seen.append(h)
            run_max = 0; cur = None; batch_max_before = 0
            for (n, s, q, h) in got:
                assert h <= run_max, "history ahead of the reads"
                run_max = max(run_max, len(s))


def test_history_runs_over_all_files(harness, tmp_path):
    """max_read_l is never reset between input files (the reference allocates its per-thread buffers once, before the
    loop over the files: src/cly_mt.c:538-556, src/cly.c:2958)"""
    a = tmp_path / "a.fq"; b = tmp_path / "b.fq"
    a.write_bytes(b"@a1\n" + b"A" * 500 + b"\n+\n" + b"5" * 500 + b"\n")
    b.write_bytes(b"@b1\nACGT\n+\n5555\n@b2\nAC\n+\n55\n")
    out = subprocess.run([harness, "2000", str(a), str(b)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")[:-1]
    assert [l.split(b"\t")[0] for l in out] == [b"a1", b"b1", b"b2"]
    assert [int(l.split(b"\t")[3]) for l in out] == [0, 500, 500]  # both reads of the second file share a batch


def bgzf_bytes(data: bytes, block=0xff00, level=6, eof_marker=True) -> bytes:
    """a BGZF file (bgzip's format: gzip members of <= 64 KB of text each, their compressed size in a 'BC' extra field)"""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for ch in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(ch) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(ch) & 0xffffffff, len(ch))
    return bytes(out)


@pytest.mark.parametrize("name", ["four", "mixed", "multi_fq", "junk", "no_nl", "crlf"])
def test_parallel_parse_path(harness, tmp_path, monkeypatch, name):
    """a wave of text is cut at guessed record starts and parsed by several threads; the result must be the
    sequential one whether the guess holds (4-line FASTQ) or the code has to fall back (anything else)"""
    data = make_inputs()[name] * 25
    path = tmp_path / "p.fq"; path.write_bytes(data)
    exp = [(n, s, q if q is not None else b"") for n, s, q in kseq_records(data)]
    monkeypatch.setenv("DSB_CLI_SEG_KB", "1")
    monkeypatch.setenv("DSB_CLI_THREADS", "7")
    monkeypatch.setenv("DSB_HARNESS_STATS", "1")
    longest = max(len(n) + 2 * len(s) for n, s, _ in exp) + 700
    for cap in (longest * 12, 60000, 1 << 23):
        p = subprocess.run([harness, str(cap), str(path)], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if name == "four":     # the guess holds on 4-line FASTQ: the pieces must actually have been parsed side by side
            assert int(re.search(rb"parallel (\d+)", p.stderr).group(1)) > 0, p.stderr
        got = run_harness(harness, cap, str(path))
        assert exp == [(n, s, q) for n, s, q, _ in got], (name, cap)
        run_max = 0
        for (n, s, q, h) in got:
            assert h <= run_max
            run_max = max(run_max, len(s))


@pytest.mark.parametrize("name", ["four", "mixed", "crlf"])
@pytest.mark.parametrize("nolib", [False, True])
def test_bgzf_and_multi_member_gzip(harness, tmp_path, monkeypatch, name, nolib):
    """BGZF input is inflated block-parallel (libdeflate if the system has it, zlib otherwise), concatenated gzip members
    are one text, bytes behind the last member are ignored (gzread's rules, src/lib/utils.c:841-905), and a BGZF file
    that turns into plain gzip half way is read to its end"""
    data = make_inputs()[name] * 9
    exp = [(n, s, q if q is not None else b"") for n, s, q in kseq_records(data)]
    if nolib:
        monkeypatch.setenv("DSB_CLI_NO_LIBDEFLATE", "1")
    monkeypatch.setenv("DSB_CLI_THREADS", "5")
    monkeypatch.setenv("DSB_CLI_SEG_KB", "1")
    half = len(data) // 2
    files = {"bgzf": bgzf_bytes(data, block=3000), "bgzf_big": bgzf_bytes(data), "members": gzip.compress(data[:half]) + gzip.compress(data[half:]),
             "trailing": gzip.compress(data) + b"\0\0garbage", "bgzf_then_gzip": bgzf_bytes(data[:half], block=2000, eof_marker=False) + gzip.compress(data[half:])}
    for kind, blob in files.items():
        path = tmp_path / (kind + ".fq.gz"); path.write_bytes(blob)
        assert gzip.decompress(blob if kind != "trailing" else blob[:-9]) == data
        for cap in (5000, 70000, 1 << 22):
            got = run_harness(harness, cap, str(path))
            assert exp == [(n, s, q) for n, s, q, _ in got], (name, kind, cap)


def test_several_gzip_files_are_inflated_ahead(harness, tmp_path, monkeypatch):
    """every upcoming .gz file has an inflater of its own; the records still arrive in file order"""
    monkeypatch.setenv("DSB_CLI_THREADS", "4")
    inputs = make_inputs(); exp = []; paths = []
    for i, name in enumerate(["four", "multi_fq", "crlf", "junk", "four", "mixed"]):
        data = inputs[name] * (i + 1)
        exp += [(n, s, q if q is not None else b"") for n, s, q in kseq_records(data)]
        p = tmp_path / ("f%d.fq%s" % (i, ".gz" if i != 2 else "")); paths.append(str(p))
        p.write_bytes(gzip.compress(data) if i != 2 else data)
    for cap in (3000, 1 << 20):
        out = subprocess.run([harness, str(cap)] + paths, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        got = [tuple(f.replace(b"\\n", b"\n").replace(b"\\r", b"\r") for f in l.split(b"\t")[:3]) for l in out.split(b"\n")[:-1]]
        assert got == exp, cap


def test_pipe_input(harness, tmp_path):
    """a FIFO (or /dev/stdin) is read through gzread like every input of the reference, plain or compressed"""
    data = make_inputs()["four"] * 3
    exp = [(n, s, q) for n, s, q in kseq_records(data)]
    for blob in (data, gzip.compress(data)):
        fifo = tmp_path / "in.fifo"
        if fifo.exists():
            fifo.unlink()
        os.mkfifo(fifo)
        w = subprocess.Popen(["sh", "-c", "cat > '%s'" % fifo], stdin=subprocess.PIPE)
        h = subprocess.Popen([harness, "4096", str(fifo)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        w.stdin.write(blob); w.stdin.close(); w.wait()
        out = h.communicate()[0]
        assert h.returncode == 0
        got = [tuple(l.split(b"\t")[:3]) for l in out.split(b"\n")[:-1]]
        assert got == exp


def test_fasta_compat_matches_the_reference_reader(harness, demo, tmp_path, monkeypatch):
    """DSB_FASTA_COMPAT=1 reproduces the reference's FASTA record loss (kseq's look-ahead character lives in each of its
    3 x 5000 kseq_t while the stream is shared, src/cly_mt.c:42-56,545-558, src/lib/utils.c:939-977): the records the
    reference binary itself reports (SAM_FULL: name and sequence of every read) on FASTA, FASTQ and mixed files, more than
    5000 records (slots used a second time), more than 10 Mbp per batch, several files (the slots outlive a file) -- are
    the records the reader delivers with the switch on.  Without it every record is delivered (the documented deviation)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "deSAMBA")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built")
    rnd = random.Random(11)
    def dna(k):
        return bytes(rnd.choice(b"ACGT") for _ in range(k))
    fa_many = b"".join(b">a%d\n%s\n" % (i, dna(rnd.randint(20, 39))) for i in range(16500))            # slots of all three workers used twice
    big = dna(600000)
    fa_long = b"".join(b">L%d\n%s\n" % (i, big[i * 1000:i * 1000 + 450000]) for i in range(60))        # batches end at 10 Mbp
    fq = b"".join(b"@q%d\n%s\n+\n%s\n" % (i, s, b"5" * len(s)) for i, s in enumerate(dna(rnd.randint(20, 39)) for _ in range(7000)))
    mixed = b"".join(b">m%d\n%s\n" % (i, dna(30)) for i in range(300)) + b"".join(b"@n%d\n%s\n+\n%s\n" % (i, dna(30), b"I" * 30) for i in range(300)) + b"".join(b">o%d\n%s\n" % (i, dna(30)) for i in range(300))
    files = []
    for name, blob in (("many.fa", fa_many), ("long.fa", fa_long), ("reads.fq", fq), ("mixed.fa", mixed), ("again.fa", fa_many[:200000].rsplit(b">", 1)[0])):
        p = tmp_path / name; p.write_bytes(blob); files.append(str(p))
    out = tmp_path / "ref.sam"
    subprocess.run([ref, "classify", "-t", "4", "-f", "SAM_FULL", demo["index"]] + files + ["-o", str(out)], check=True, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    exp = []
    for ln in open(out, "rb"):
        f = ln.rstrip(b"\t\n").split(b"\t")
        if not (int(f[1]) & 0x900):          # one line per read: unmapped or primary
            exp.append((f[0], f[9]))
    n_all = sum(b.count(b">") + (b.count(b"\n@")) + b.startswith(b"@") for b in (open(f, "rb").read() for f in files))
    assert 0 < len(exp) < n_all              # the reference did lose records
    monkeypatch.setenv("DSB_FASTA_COMPAT", "1")
    for cap in (1 << 16, 1 << 24):
        o = subprocess.run([harness, str(cap)] + files, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        got = [tuple(l.split(b"\t")[:2]) for l in o.split(b"\n")[:-1]]
        # reads the reference reverse-complements print their sequence reversed: compare the forward sequence or its reverse complement
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        assert len(got) == len(exp), (cap, len(got), len(exp))
        for (gn, gs), (en, es) in zip(got, exp):
            assert gn == en and (gs == es or gs.translate(comp)[::-1] == es), (cap, gn, en)
    monkeypatch.delenv("DSB_FASTA_COMPAT")
    o = subprocess.run([harness, str(1 << 24)] + files, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert o.count(b"\n") == n_all


def _fq(prefix, n):
    return b"".join(b"@%s%d\nACGTACGTAC\n+\n5555555555\n" % (prefix, i) for i in range(n))


def test_files_are_opened_one_window_at_a_time(harness, tmp_path):
    """more input files than the descriptor limit allows open at once (plain and gzip mixed): the reference opens one file at a
    time (src/cly_mt.c:551-558); the reader here opens a file only when its inflater starts or at its turn"""
    import resource
    paths = []; exp = []
    for k in range(40):
        data = _fq(b"f%d_" % k, 3)
        p = tmp_path / ("f%02d.fq%s" % (k, ".gz" if k % 3 == 1 else ""))
        if k % 3 == 1:
            with gzip.open(p, "wb") as f:
                f.write(data)
        else:
            p.write_bytes(data)
        paths.append(str(p)); exp += [n for n, _, _ in kseq_records(data)]
    def low_limit():
        resource.setrlimit(resource.RLIMIT_NOFILE, (16, 16))
    out = subprocess.run([harness, "4096"] + paths, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, preexec_fn=low_limit)
    assert [l.split(b"\t")[0] for l in out.stdout.split(b"\n")[:-1]] == exp


def test_fifos_fed_one_after_the_other(harness, tmp_path):
    """several named pipes written by ONE writer, one after the other: the reader must not open the second before the first is
    read to its end (open() of a FIFO blocks until its writer comes, and the writer is busy with the first)"""
    import threading
    fifos = [str(tmp_path / ("p%d" % i)) for i in range(3)]
    for f in fifos:
        os.mkfifo(f)
    datas = [_fq(b"a", 4000), gzip.compress(_fq(b"b", 3000)), _fq(b"c", 10)]          # (> 64 KB: more than a pipe holds)
    def writer():
        for f, d in zip(fifos, datas):
            with open(f, "wb") as h:
                h.write(d)
    th = threading.Thread(target=writer, daemon=True); th.start()
    out = subprocess.run([harness, "65536"] + fifos, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60).stdout
    th.join(10)
    names = [l.split(b"\t")[0] for l in out.split(b"\n")[:-1]]
    assert names == [b"a%d" % i for i in range(4000)] + [b"b%d" % i for i in range(3000)] + [b"c%d" % i for i in range(10)]


def test_dash_is_stdin(harness, tmp_path):
    """`-` names the standard input (xzopen, src/lib/utils.c:64-68), plain or gzip"""
    plain = _fq(b"s", 50); other = tmp_path / "o.fq"; other.write_bytes(_fq(b"o", 5))
    for data in (plain, gzip.compress(plain)):
        out = subprocess.run([harness, "4096", str(other), "-"], input=data, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60).stdout
        assert [l.split(b"\t")[0] for l in out.split(b"\n")[:-1]] == [b"o%d" % i for i in range(5)] + [b"s%d" % i for i in range(50)]
