"""desamba_amd -- MI355X-native `deSAMBA classify` hot path.

Thin ctypes binding over the C-ABI in include/desamba_amd.h (libdesamba_amd.so, built in-tree by
__graft_entry__.build()).  All compute runs in HIP kernels on gfx950; there is no CPU fallback:
loading fails loudly when the library is missing and Ctx() fails when no gfx950 device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSB_LIB_PATH", os.path.join(_HERE, "libdesamba_amd.so"))   # override: A/B builds in experiments

DSB_OK, DSB_EIO, DSB_ENODEV, DSB_ENOMEM, DSB_EINVAL, DSB_ECAP = 0, -1, -2, -3, -4, -5


class DsbOpts(C.Structure):
    _fields_ = [("L_min_matching", C.c_int), ("min_score", C.c_int), ("max_sec_N", C.c_int), ("n_slots", C.c_int),
                ("max_read_len", C.c_uint32), ("max_batch_reads", C.c_uint32), ("input_slots", C.c_int), ("reserved", C.c_int), ("max_batch_bases", C.c_uint64)]


class DsbRead(C.Structure):
    _fields_ = [("name", C.c_char_p), ("seq", C.c_char_p), ("qual", C.c_char_p), ("len", C.c_uint32)]


class DsbHit(C.Structure):
    _fields_ = [("ref_ID", C.c_uint32), ("t_st", C.c_uint32), ("t_ed", C.c_uint32), ("q_st", C.c_uint32), ("q_ed", C.c_uint32),
                ("sum_score", C.c_uint32), ("indel", C.c_uint32), ("direction", C.c_uint8), ("primary", C.c_uint8),
                ("pri_index", C.c_uint8), ("pad", C.c_uint8)]

    def key(self):
        return (self.ref_ID, self.t_st, self.t_ed, self.q_st, self.q_ed, self.sum_score, self.direction, self.primary, self.pri_index)


class DsbReadResult(C.Structure):
    _fields_ = [("first", C.c_uint32), ("n", C.c_uint32), ("status", C.c_int32), ("fast", C.c_uint32), ("device_us", C.c_uint32), ("n_anc", C.c_uint32)]


class DsbResult(C.Structure):
    _fields_ = [("reads", C.POINTER(DsbReadResult)), ("hits", C.POINTER(DsbHit)), ("n_hits", C.c_size_t)]


class DsbSeed(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("len", C.c_uint32), ("top", C.c_uint8), ("pad", C.c_uint8 * 3)]


class DsbTiming(C.Structure):
    _fields_ = [("encode_ms", C.c_float), ("seed_probe_ms", C.c_float), ("classify_ms", C.c_float), ("total_ms", C.c_float),
                ("windows", C.c_uint64), ("probes_t1", C.c_uint64), ("bases", C.c_uint64),
                ("order_ms", C.c_float), ("tail_ms", C.c_float), ("n_early", C.c_uint32), ("n_retry", C.c_uint32),
                ("n_regrow", C.c_uint32), ("seed_scan", C.c_uint32),
                ("n_occ", C.c_uint64), ("n_mem", C.c_uint64), ("n_sa", C.c_uint64), ("ref_bases", C.c_uint64),
                ("main_occ", C.c_uint64), ("main_mem", C.c_uint64), ("main_sa", C.c_uint64), ("main_ref_bases", C.c_uint64),
                ("n_heavy_mw", C.c_uint32), ("n_requeue", C.c_uint32), ("upload_bytes", C.c_uint64)]


class DsbBuildStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_bases", "n_refs", "n_kmer", "n_unitig", "n_rows")] + \
               [(n, C.c_double) for n in ("parse_s", "sort_s", "graph_s", "walk_s", "rows_s", "tables_s", "write_s", "total_s")] + \
               [(n, C.c_uint64) for n in ("budget_bytes", "peak_device_bytes")] + \
               [(n, C.c_uint32) for n in ("ranges_kmers", "ranges_unitig_numbers", "ranges_rows", "ranges_exist")] + \
               [("spilled_bytes", C.c_uint64)]


class DsbChunk(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("hist_max_before", C.c_uint32), ("rank", C.c_int32)]


EXPORTS = ["dsb_index_open", "dsb_index_close", "dsb_index_n_ref", "dsb_index_ref_name", "dsb_index_ref_len", "dsb_index_ek_len",
           "dsb_index_occ_host", "dsb_ctx_create", "dsb_ctx_destroy", "dsb_ctx_reset_history", "dsb_ctx_reload_env", "dsb_classify_batch",
           "dsb_batch_upload", "dsb_batch_upload_fastq", "dsb_batch_upload_text", "dsb_ctx_set_history", "dsb_host_alloc", "dsb_host_free", "dsb_host_cpus", "dsb_batch_run", "dsb_batch_fetch", "dsb_batch_timing", "dsb_batch_seeds", "dsb_batch_exist_bits",
           "dsb_format_sam", "dsb_format_des", "dsb_strerror", "dsb_version",
           "dsb_device_count", "dsb_ctx_select_slot", "dsb_ctx_create_multi", "dsb_multi_destroy", "dsb_multi_n", "dsb_multi_ctx",
           "dsb_multi_reset_history", "dsb_multi_last_calls", "dsb_multi_classify_batch", "dsb_shard_plan", "dsb_ctx_use_synthetic_filter", "dsb_synthetic_filter_bit", "dsb_index_prefix_interval", "dsb_index_build"]

_lib = None


def lib():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("desamba_amd: %s is missing -- run __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.dsb_index_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.dsb_index_close.argtypes = [C.c_void_p]
    L.dsb_index_n_ref.argtypes = [C.c_void_p]; L.dsb_index_n_ref.restype = C.c_uint64
    L.dsb_index_ref_name.argtypes = [C.c_void_p, C.c_uint32]; L.dsb_index_ref_name.restype = C.c_char_p
    L.dsb_index_ref_len.argtypes = [C.c_void_p, C.c_uint32]; L.dsb_index_ref_len.restype = C.c_uint64
    L.dsb_index_ek_len.argtypes = [C.c_void_p]
    L.dsb_index_occ_host.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint8)]; L.dsb_index_occ_host.restype = C.c_uint64
    L.dsb_ctx_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(DsbOpts), C.POINTER(C.c_void_p)]
    L.dsb_ctx_destroy.argtypes = [C.c_void_p]
    L.dsb_ctx_reset_history.argtypes = [C.c_void_p]
    L.dsb_ctx_reload_env.argtypes = [C.c_void_p]; L.dsb_ctx_reload_env.restype = None
    L.dsb_classify_batch.argtypes = [C.c_void_p, C.POINTER(DsbRead), C.c_size_t, C.POINTER(DsbResult)]
    L.dsb_batch_upload.argtypes = [C.c_void_p, C.POINTER(DsbRead), C.c_size_t]
    L.dsb_batch_upload_fastq.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t]; L.dsb_batch_upload_fastq.restype = C.c_long
    L.dsb_batch_run.argtypes = [C.c_void_p]
    L.dsb_batch_fetch.argtypes = [C.c_void_p, C.POINTER(DsbResult)]
    L.dsb_batch_timing.argtypes = [C.c_void_p, C.POINTER(DsbTiming)]
    L.dsb_batch_seeds.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(DsbSeed), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.dsb_batch_exist_bits.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint32)]
    L.dsb_format_sam.argtypes = [C.c_void_p, C.POINTER(DsbRead), C.POINTER(DsbHit), C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.dsb_format_sam.restype = C.c_long
    L.dsb_format_des.argtypes = [C.c_void_p, C.POINTER(DsbRead), C.POINTER(DsbReadResult), C.POINTER(DsbHit), C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.dsb_format_des.restype = C.c_long
    L.dsb_batch_upload_text.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t]
    L.dsb_ctx_set_history.argtypes = [C.c_void_p, C.c_uint32]; L.dsb_ctx_set_history.restype = None
    L.dsb_host_alloc.argtypes = [C.c_size_t]; L.dsb_host_alloc.restype = C.c_void_p
    L.dsb_host_free.argtypes = [C.c_void_p]; L.dsb_host_free.restype = None
    L.dsb_device_count.argtypes = []; L.dsb_device_count.restype = C.c_int
    L.dsb_ctx_select_slot.argtypes = [C.c_void_p, C.c_int]
    L.dsb_ctx_create_multi.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(DsbOpts), C.POINTER(C.c_void_p)]
    L.dsb_multi_destroy.argtypes = [C.c_void_p]; L.dsb_multi_destroy.restype = None
    L.dsb_multi_n.argtypes = [C.c_void_p]
    L.dsb_multi_ctx.argtypes = [C.c_void_p, C.c_int]; L.dsb_multi_ctx.restype = C.c_void_p
    L.dsb_multi_reset_history.argtypes = [C.c_void_p]; L.dsb_multi_reset_history.restype = None
    L.dsb_multi_last_calls.argtypes = [C.c_void_p, C.c_int]; L.dsb_multi_last_calls.restype = C.c_uint32
    L.dsb_multi_classify_batch.argtypes = [C.c_void_p, C.POINTER(DsbRead), C.c_size_t, C.POINTER(DsbResult)]
    L.dsb_shard_plan.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_uint64, C.c_uint32, C.POINTER(DsbChunk), C.c_size_t, C.POINTER(C.c_size_t)]
    L.dsb_ctx_use_synthetic_filter.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
    L.dsb_synthetic_filter_bit.argtypes = [C.c_int, C.c_uint64, C.c_double]
    L.dsb_index_prefix_interval.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.dsb_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(DsbBuildStats)]
    L.dsb_index_close.restype = None; L.dsb_ctx_destroy.restype = None; L.dsb_ctx_reset_history.restype = None
    L.dsb_strerror.argtypes = [C.c_int]; L.dsb_strerror.restype = C.c_char_p
    L.dsb_version.restype = C.c_char_p
    _lib = L
    return L


class DsbError(RuntimeError):
    def __init__(self, code, what):
        RuntimeError.__init__(self, "%s: %s (%d)" % (what, lib().dsb_strerror(code).decode(), code))
        self.code = code


class Index:
    """load_idx (src/idx.c:1103): the ten deSAMBA.* files of an index directory."""

    def __init__(self, path):
        self.h = C.c_void_p()
        rc = lib().dsb_index_open(os.fsencode(path), C.byref(self.h))
        if rc != 0:
            raise DsbError(rc, "dsb_index_open(%s)" % path)
        self.path = path

    def close(self):
        if self.h:
            lib().dsb_index_close(self.h); self.h = C.c_void_p()

    @property
    def n_ref(self):
        return lib().dsb_index_n_ref(self.h)

    @property
    def ek_len(self):
        return lib().dsb_index_ek_len(self.h)

    def ref_name(self, i):
        return lib().dsb_index_ref_name(self.h, i).decode()

    def occ_host(self, r, c):
        cc = C.c_uint8(c)
        v = lib().dsb_index_occ_host(self.h, r, C.byref(cc))
        return v, cc.value


def build_index(fasta, out_dir, kmer_srt=None, device=0):
    """`deSAMBA index` on the GPU (build_index_main, src/idx.c:1254-1282): writes the deSAMBA.* files of an index
    directory from a reference FASTA; kmer_srt=None enumerates the 31-mers from the FASTA itself.  Returns DsbBuildStats."""
    st = DsbBuildStats()
    rc = lib().dsb_index_build(kmer_srt.encode() if kmer_srt else None, fasta.encode(), out_dir.encode(), device, C.byref(st))
    if rc != 0:
        raise DsbError(rc, "dsb_index_build(%s)" % fasta)
    return st


def make_reads(records):
    """records: list of (name, seq, qual) of bytes/str -> ctypes array (keeps the buffers alive)."""
    n = len(records)
    arr = (DsbRead * n)()
    keep = []
    for i, (name, seq, qual) in enumerate(records):
        name = name if isinstance(name, bytes) else name.encode()
        seq = seq if isinstance(seq, bytes) else seq.encode()
        qual = (qual if isinstance(qual, bytes) else qual.encode()) if qual is not None else None
        keep.append((name, seq, qual))
        arr[i].name, arr[i].seq, arr[i].qual, arr[i].len = name, seq, qual, len(seq)
    arr._keep = keep
    return arr


class Ctx:
    """classify_main's set-up (src/cly_mt.c:518-550) on one GPU."""

    def __init__(self, index, device=0, L_min_matching=170, min_score=64, max_sec_N=5, n_slots=0, max_read_len=0, max_batch_reads=0, input_slots=1, max_batch_bases=0):
        self.index = index
        self.opts = DsbOpts(L_min_matching, min_score, max_sec_N, n_slots, max_read_len, max_batch_reads, input_slots, 0, max_batch_bases)
        self.h = C.c_void_p()
        rc = lib().dsb_ctx_create(index.h, device, C.byref(self.opts), C.byref(self.h))
        if rc != 0:
            raise DsbError(rc, "dsb_ctx_create(device %d)" % device)
        self.reads = None

    def close(self):
        if self.h:
            lib().dsb_ctx_destroy(self.h); self.h = C.c_void_p()

    def reset_history(self):
        lib().dsb_ctx_reset_history(self.h)

    def reload_env(self):
        """the DSB_* switches are read once, when the ctx is made: read them again (tests that change them on a living ctx)"""
        lib().dsb_ctx_reload_env(self.h)

    def set_history(self, max_len_before):
        lib().dsb_ctx_set_history(self.h, max_len_before)

    def use_synthetic_filter(self, table_bytes, fill):
        """measurement hook: synthetic exist-kmer tables for this ctx (seed lookup only); upload the batch afterwards"""
        rc = lib().dsb_ctx_use_synthetic_filter(self.h, table_bytes, fill)
        if rc != 0:
            raise DsbError(rc, "dsb_ctx_use_synthetic_filter")
        self.synthetic = (table_bytes, fill)

    def select_slot(self, slot):
        rc = lib().dsb_ctx_select_slot(self.h, slot)
        if rc != 0:
            raise DsbError(rc, "dsb_ctx_select_slot(%d)" % slot)

    def upload_text(self, text_ptr, text_len, seq_off, seq_len, n):
        """sequences inside one host blob (pinned if it came from dsb_host_alloc): one H2D copy, no per-read gather"""
        self.reads = None
        rc = lib().dsb_batch_upload_text(self.h, text_ptr, text_len, seq_off, seq_len, n)
        if rc != 0:
            raise DsbError(rc, "dsb_batch_upload_text")

    def upload(self, reads):
        self.reads = reads
        rc = lib().dsb_batch_upload(self.h, reads, len(reads))
        if rc != 0:
            raise DsbError(rc, "dsb_batch_upload")

    def upload_fastq(self, path, skip=0, max_reads=1 << 62):
        """stage a plain-text FASTQ file straight into HBM; returns the number of reads"""
        self.reads = None
        n = lib().dsb_batch_upload_fastq(self.h, os.fsencode(path), skip, max_reads)
        if n < 0:
            raise DsbError(int(n), "dsb_batch_upload_fastq(%s)" % path)
        self.n_uploaded = int(n)
        return int(n)

    def run(self):
        rc = lib().dsb_batch_run(self.h)
        if rc != 0:
            raise DsbError(rc, "dsb_batch_run")

    def fetch(self, strict=True):
        res = DsbResult()
        rc = lib().dsb_batch_fetch(self.h, C.byref(res))
        if rc != 0 and (strict or rc != DSB_ECAP):
            raise DsbError(rc, "dsb_batch_fetch")
        return res

    def classify(self, reads, strict=True):
        self.upload(reads); self.run()
        return self.fetch(strict)

    def timing(self):
        t = DsbTiming()
        lib().dsb_batch_timing(self.h, C.byref(t))
        return t

    def seeds(self, read, strand):
        cap = (self.reads[read].len >> 1) + 64
        buf = (DsbSeed * cap)(); n = C.c_uint32(); ts = C.c_uint32()
        rc = lib().dsb_batch_seeds(self.h, read, strand, buf, cap, C.byref(n), C.byref(ts))
        if rc != 0:
            raise DsbError(rc, "dsb_batch_seeds")
        return [(buf[i].offset, buf[i].len, buf[i].top) for i in range(n.value)], ts.value

    def exist_bits(self, read, strand):
        cap = self.reads[read].len + 1
        buf = (C.c_uint8 * cap)(); n = C.c_uint32()
        rc = lib().dsb_batch_exist_bits(self.h, read, strand, buf, cap, C.byref(n))
        if rc != 0:
            raise DsbError(rc, "dsb_batch_exist_bits")
        return bytes(buf[:n.value])

    def sam(self, res, full=False, reads=None):
        """Format a whole batch exactly as output_one_result_sam does (src/cly_mt.c:245-344)."""
        return format_sam(self.index, reads if reads is not None else self.reads, res, self.opts.max_sec_N, full)


def format_sam(index, reads, res, max_sec_N=5, full=False):
    out = []
    buf = C.create_string_buffer(1 << 20)
    for i in range(len(reads)):
        rr = res.reads[i]
        hits = C.cast(C.byref(res.hits.contents, rr.first * C.sizeof(DsbHit)), C.POINTER(DsbHit)) if rr.n else None
        cap = len(buf)
        need = 4096 + 700 * rr.n + (2 * reads[i].len if full else 0)
        if need > cap:
            buf = C.create_string_buffer(need)
        n = lib().dsb_format_sam(index.h, C.byref(reads[i]), hits, rr.n, max_sec_N, 1 if full else 0, buf, len(buf))
        if n < 0:
            raise DsbError(DSB_EINVAL, "dsb_format_sam")
        out.append(buf.raw[:n])
    return b"".join(out)


class Multi:
    """dsb_ctx_create_multi: one context per listed device; classify() shards a batch over them (no collective)."""

    def __init__(self, index, devices, L_min_matching=170, min_score=64, max_sec_N=5, n_slots=0):
        self.index = index
        self.opts = DsbOpts(L_min_matching, min_score, max_sec_N, n_slots, 0, 0, 1, 0)
        ids = (C.c_int * len(devices))(*devices)
        self.h = C.c_void_p()
        rc = lib().dsb_ctx_create_multi(index.h, ids, len(devices), C.byref(self.opts), C.byref(self.h))
        if rc != 0:
            raise DsbError(rc, "dsb_ctx_create_multi(%r)" % (devices,))

    def close(self):
        if self.h:
            lib().dsb_multi_destroy(self.h); self.h = C.c_void_p()

    def reset_history(self):
        lib().dsb_multi_reset_history(self.h)

    def last_calls(self):
        """dsb_classify_batch calls each context made in the last classify()"""
        return [int(lib().dsb_multi_last_calls(self.h, i)) for i in range(int(lib().dsb_multi_n(self.h)))]

    def classify(self, reads, strict=True):
        res = DsbResult()
        rc = lib().dsb_multi_classify_batch(self.h, reads, len(reads), C.byref(res))
        if rc != 0 and (strict or rc != DSB_ECAP):
            raise DsbError(rc, "dsb_multi_classify_batch")
        return res


def shard_plan(lengths, world, chunk_bases=0, chunk_reads=0):
    """dsb_shard_plan -> list of (start, end, hist_max_before, rank)"""
    n = len(lengths)
    arr = (C.c_uint32 * max(n, 1))(*lengths)
    cnt = C.c_size_t()
    lib().dsb_shard_plan(arr, n, world, chunk_bases, chunk_reads, None, 0, C.byref(cnt))
    out = (DsbChunk * max(cnt.value, 1))()
    rc = lib().dsb_shard_plan(arr, n, world, chunk_bases, chunk_reads, out, cnt.value, C.byref(cnt))
    if rc != 0:
        raise DsbError(rc, "dsb_shard_plan")
    return [(out[i].start, out[i].end, out[i].hist_max_before, out[i].rank) for i in range(cnt.value)]


def read_fastq(path, limit=None):
    """Plain-text FASTQ/FASTA reader with the record rules of the reference's kseq_read (src/lib/utils.c:939-977; the
    OLD kseq: '\\r' stays in sequence and quality, the first character of a sequence line is data even if it is '\\n',
    quality is read in whole lines).  Returns (name, seq, qual) tuples; records with a quality string of the wrong
    length are dropped as read_reads (src/cly_mt.c:42-56) drops them."""
    with open(path, "rb") as f:
        data = f.read()
    return parse_fastq(data, limit)


def parse_fastq(data, limit=None):
    recs = []
    n = len(data)
    p = 0
    last = 0
    space = b" \t\n\v\f\r"
    while True:
        if last == 0:
            a, b = data.find(b">", p), data.find(b"@", p)
            if a < 0 and b < 0:
                break
            p = min(x for x in (a, b) if x >= 0) + 1
        q = p
        while q < n and data[q] not in space:
            q += 1
        if q >= n and q == p:
            break
        name = data[p:q]
        if q < n and data[q:q + 1] != b"\n":
            e = data.find(b"\n", q)
            q = n if e < 0 else e
        p = min(q + 1, n) if q < n else n
        seq = []
        c = -1
        while True:
            if p >= n:
                c = -1
                break
            c = data[p]
            if c in b">+@":
                p += 1
                break
            e = data.find(b"\n", p + 1)
            e = n if e < 0 else e
            seq.append(data[p:e])
            p = min(e + 1, n)
        seq = b"".join(seq)
        last = c if c in (ord(">"), ord("@")) else 0
        qual = None
        bad = False
        if c == ord("+"):
            e = data.find(b"\n", p)
            if e < 0:
                break
            p = e + 1
            ql = []
            tot = 0
            while True:
                if p >= n:
                    break
                e = data.find(b"\n", p)
                e = n if e < 0 else e
                ql.append(data[p:e]); tot += e - p
                p = min(e + 1, n)
                if tot >= len(seq):
                    break
            qual = b"".join(ql)
            last = 0
            bad = len(qual) != len(seq)
        if not bad:
            recs.append((name, seq, qual))
            if limit and len(recs) >= limit:
                break
    return recs
