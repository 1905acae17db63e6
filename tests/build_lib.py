"""Test helpers for index construction: the host emulation of the builder stages (tests/emu/emu_build.cpp), digests of an
index directory with the reference's two indeterminacies removed, and a plain-Python restatement of the reference's
FASTA reader used to make k-mer lists for awkward inputs."""
import ctypes as C
import hashlib
import os
import struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTS = [".acg", ".bwt", ".exk0", ".exk1", ".exki", ".ref_b", ".ref_i", ".ref_p", ".sa", ".unv"]
_emu = None


def _load_emu():
    global _emu
    if _emu is None:
        p = os.path.join(ROOT, "tests", "emu", "libdsbemu_build.so")
        if not os.path.exists(p):
            raise RuntimeError("%s missing -- run __graft_entry__.build()" % p)
        _emu = C.CDLL(p)
        _emu.dsb_emu_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]
        _emu.dsb_emu_index_build_parts.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]


def emu_build(fasta, out_dir, kmer_srt=None):
    """run the builder stages on the host; returns (n_kmer, n_unitig, n_rows, n_refs)"""
    _load_emu()
    st = (C.c_uint64 * 4)()
    rc = _emu.dsb_emu_index_build(kmer_srt.encode() if kmer_srt else None, fasta.encode(), out_dir.encode(), st)
    if rc:
        raise RuntimeError("dsb_emu_index_build(%s) = %d" % (fasta, rc))
    return tuple(st)


def emu_build_parts(fasta, out_dir, kmer_srt=None, budget=1 << 40, parts=0):
    """the same through dsb_build_run_parts (dsb_build_parts.h): `parts` ranges of prefixes per stage, or as many as `budget` bytes ask
    for; returns a dict with the counts, the peak bytes the backend held and the number of ranges of each stage"""
    _load_emu()
    st = (C.c_uint64 * 11)()
    rc = _emu.dsb_emu_index_build_parts(kmer_srt.encode() if kmer_srt else None, fasta.encode(), out_dir.encode(), budget, parts, st)
    if rc:
        raise RuntimeError("dsb_emu_index_build_parts(%s) = %d" % (fasta, rc))
    return dict(zip(("n_kmer", "n_unitig", "n_rows", "n_refs", "peak", "parts_kmers", "parts_uid", "parts_rows", "start_windows", "parts_exist", "spilled_bytes"), st))


def n_rows_of(d):
    unv = open(os.path.join(d, "deSAMBA.unv"), "rb").read()
    n = struct.unpack_from("<Q", unv)[0]
    return sum(struct.unpack_from("<II", unv, 8 + 8 * i)[1] + 1 for i in range(n - 1))


def canonical_bytes(d, e):
    """bytes of index file e of directory d, minus what the reference leaves undefined: the bytes behind the last BWT symbol
    when the index has fewer than 257 blocks (uninitialised heap, src/bwt.c:222-238) and the padding behind each reference
    name (.ref_i, kv_pushp'd and strcpy'd, src/idx.c:586-589)"""
    b = bytearray(open(os.path.join(d, "deSAMBA" + e), "rb").read())
    if e == ".bwt":
        rows = n_rows_of(d)
        n_blk = (rows + 255) // 256; n_bin = (rows + 1) // 2
        if n_blk <= 256 and n_bin % 128:
            v = n_bin - (n_blk - 1) * 128
            lo = 8 + (n_blk - 1) * 168 + 40 + v
            b[lo:8 + n_blk * 168] = bytes(8 + n_blk * 168 - lo)
    if e == ".ref_i":
        n = struct.unpack_from("<Q", b)[0]
        b = bytearray(b"".join(bytes(b[8 + 144 * i:8 + 144 * i + 128]).split(b"\0")[0] + b"\0" + bytes(b[8 + 144 * i + 128:8 + 144 * (i + 1)]) for i in range(n)))
    return bytes(b)


def digest_dir(d):
    out = {e: hashlib.md5(canonical_bytes(d, e)).hexdigest() for e in EXTS}
    out["n_rows"] = n_rows_of(d)
    return out


def reader_view(text):
    """records (name, sequence bytes) as kseq_read (src/lib/utils.c:939-977) delivers them"""
    recs = []; p = 0; n = len(text); last = 0
    while True:
        if last == 0:
            while p < n and text[p] not in b">@":
                p += 1
            if p >= n:
                break
            p += 1
        if p >= n:
            break
        q = p
        while q < n and not chr(text[q]).isspace():
            q += 1
        name = text[p:q]
        c = text[q] if q < n else -1
        p = q + 1
        if c != -1 and c != 10:
            e = text.find(b"\n", p); p = n if e < 0 else e + 1
        seq = bytearray(); c = -1
        while p < n:
            c = text[p]; p += 1
            if c in (62, 43, 64):
                break
            seq.append(c)
            e = text.find(b"\n", p)
            if e < 0:
                seq += text[p:]; p = n
            else:
                seq += text[p:e]; p = e + 1
            c = -1
        last = c if c in (62, 64) else 0
        if c == ord("+"):
            e = text.find(b"\n", p)
            if e < 0:
                break
            p = e + 1; ql = 0
            while p < n:
                e = text.find(b"\n", p)
                if e < 0:
                    ql += n - p; p = n
                else:
                    ql += e - p; p = e + 1
                if ql >= len(seq):
                    break
            last = 0
            if ql != len(seq):
                break
        recs.append((bytes(name), bytes(seq)))
        if p >= n and last == 0:
            break
    return recs


def write_kmer_srt_from_text(recs, path):
    """kmer.srt (u64 n + sorted distinct 31-mers of the ACGT runs) of a list of records; small inputs only"""
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    ks = set(); mask = (1 << 62) - 1
    for _, s in recs:
        key = 0; run = 0
        for ch in s:
            b = code.get(ch)
            if b is None:
                run = 0; continue
            key = ((key << 2) | b) & mask; run += 1
            if run >= 31:
                ks.add(key)
    ks = sorted(ks)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(ks))); f.write(struct.pack("<%dQ" % len(ks), *ks))
