"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by desamba_amd/."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")


class OraHit(C.Structure):
    _fields_ = [("ref_ID", C.c_uint32), ("t_st", C.c_uint32), ("t_ed", C.c_uint32), ("q_st", C.c_uint32), ("q_ed", C.c_uint32),
                ("sum_score", C.c_uint32), ("indel", C.c_uint32), ("direction", C.c_uint8), ("primary", C.c_uint8),
                ("pri_index", C.c_uint8), ("pad", C.c_uint8)]

    def key(self):
        return (self.ref_ID, self.t_st, self.t_ed, self.q_st, self.q_ed, self.sum_score, self.direction, self.primary, self.pri_index)


class OraSeed(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("len", C.c_uint32), ("top", C.c_uint8)]


class OraIdx(C.Structure):
    _fields_ = [("blob", C.c_uint8 * 16384)]   # opaque: sizeof(ora_idx_t) < 16 KiB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_SO)
        L.ora_idx_load.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.ora_ctx_new.restype = C.c_void_p
        L.ora_ctx_free.argtypes = [C.c_void_p]
        L.ora_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.POINTER(OraHit))]
        L.ora_ctx_set_history.argtypes = [C.c_void_p, C.c_int]
        L.ora_ctx_reset_history.argtypes = [C.c_void_p]
        L.ora_last_seeds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(OraSeed)), C.POINTER(C.c_uint32)]
        L.ora_exist_bits.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint8)]
        L.ora_last_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.ora_last_anchors.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ora_occ.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint8)]; L.ora_occ.restype = C.c_uint64
        L.ora_classify_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.ora_classify_file.restype = C.c_long
        _lib = L
    return _lib


class Oracle:
    def __init__(self, index_dir, min_len=170, min_score=64):
        self.idx = OraIdx()
        rc = lib().ora_idx_load(C.byref(self.idx), os.fsencode(index_dir), min_len, min_score)
        if rc != 0:
            raise RuntimeError("ora_idx_load(%s) = %d" % (index_dir, rc))
        self.ctx = lib().ora_ctx_new()

    def classify(self, seq, hist_max=None):
        if hist_max is not None:
            lib().ora_ctx_set_history(self.ctx, hist_max)
        hp = C.POINTER(OraHit)()
        n = lib().ora_classify(self.ctx, C.byref(self.idx), seq, len(seq), C.byref(hp))
        return [hp[i].key() for i in range(n)]

    def seeds(self, strand):
        sp = C.POINTER(OraSeed)(); ts = C.c_uint32()
        n = lib().ora_last_seeds(self.ctx, strand, C.byref(sp), C.byref(ts))
        return [(sp[i].offset, sp[i].len, sp[i].top) for i in range(n)], ts.value

    def exist_bits(self, seq, strand):
        k = 16
        out = (C.c_uint8 * (len(seq) + 1))()
        lib().ora_exist_bits(C.byref(self.idx), seq, len(seq), strand, out)
        return out

    def n_anc(self):
        """anchor_v.n of the last read (what the DES header prints)"""
        return int(lib().ora_last_anchors(self.ctx, None, 0))

    def counters(self):
        out = (C.c_uint64 * 8)()
        lib().ora_last_counters(self.ctx, out)
        return list(out)

    def occ(self, r, c):
        cc = C.c_uint8(c)
        v = lib().ora_occ(C.byref(self.idx), r, C.byref(cc))
        return v, cc.value

    def classify_file(self, reads, out, max_sec=5, full=0, threads=1):
        nb = C.c_uint64()
        n = lib().ora_classify_file(C.byref(self.idx), os.fsencode(reads), os.fsencode(out), max_sec, full, threads, C.byref(nb))
        return n, nb.value
