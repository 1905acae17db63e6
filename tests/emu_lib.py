"""ctypes binding of the host emulation of the device code (tests/emu/libdsbemu.so). TEST ONLY."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.environ.get("DSB_EMU_LIB", os.path.join(ROOT, "tests", "emu", "libdsbemu.so"))   # override: the sanitizer build of tests/tools/emu_sanitize.sh


class EmuHit(C.Structure):
    _fields_ = [("ref_ID", C.c_uint32), ("t_st", C.c_uint32), ("t_ed", C.c_uint32), ("q_st", C.c_uint32), ("q_ed", C.c_uint32),
                ("sum_score", C.c_uint32), ("indel", C.c_uint32), ("direction", C.c_uint8), ("primary", C.c_uint8),
                ("pri_index", C.c_uint8), ("pad", C.c_uint8)]

    def key(self):
        return (self.ref_ID, self.t_st, self.t_ed, self.q_st, self.q_ed, self.sum_score, self.direction, self.primary, self.pri_index)


class EmuSeed(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("len", C.c_uint16), ("top", C.c_uint16)]


class Emu:
    def __init__(self, index_dir, min_len=170, min_score=64):
        L = C.CDLL(EMU_SO)
        L.dsb_index_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.emu_new.argtypes = [C.c_void_p, C.c_int, C.c_int]; L.emu_new.restype = C.c_void_p
        L.emu_classify.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(EmuHit), C.c_int, C.c_void_p, C.c_void_p]
        L.emu_n_anc.argtypes = [C.c_void_p]; L.emu_n_anc.restype = C.c_uint32
        L.emu_sms_peak.argtypes = [C.c_void_p]; L.emu_sms_peak.restype = C.c_uint32
        L.emu_seeds.argtypes = [C.c_void_p, C.c_int, C.POINTER(EmuSeed), C.c_int, C.POINTER(C.c_uint32)]
        L.emu_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        if hasattr(L, "dsb_emu_findings"):
            L.dsb_emu_findings.argtypes = [C.c_char_p, C.c_size_t]; L.dsb_emu_findings.restype = C.c_int
        L.emu_scan_seeds.argtypes = [C.c_void_p, C.c_int, C.POINTER(EmuSeed), C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self.L = L
        self.idx = C.c_void_p()
        rc = L.dsb_index_open(os.fsencode(index_dir), C.byref(self.idx))
        if rc:
            raise RuntimeError("dsb_index_open %d" % rc)
        self.e = L.emu_new(self.idx, min_len, min_score)
        self.buf = (EmuHit * 512)()

    def classify(self, seq, hist_max=0, want_bits=False):
        bF = bR = None
        if want_bits:
            bF = (C.c_uint8 * (len(seq) + 1))(); bR = (C.c_uint8 * (len(seq) + 1))()
        n = self.L.emu_classify(self.e, seq, len(seq), hist_max, self.buf, 512, bF, bR)
        if n < 0:
            raise RuntimeError("device-code status %#x" % (-n))
        hits = [self.buf[i].key() for i in range(min(n, 512))]
        return (hits, bF, bR) if want_bits else hits

    def findings(self):
        """64-lane emulation (tests/emu/emu_simt.cpp): what the race detector found since the last call -> list of text lines"""
        if not hasattr(self.L, "dsb_emu_findings"):
            return []
        buf = C.create_string_buffer(1 << 16)
        self.L.dsb_emu_findings(buf, len(buf))
        return [l for l in buf.value.decode().split("\n") if l]

    def n_anc(self):
        return int(self.L.emu_n_anc(self.e))

    def counters(self):
        """work counters of the last read: (occ, MEM searches, SA lookups, reference bases fetched)"""
        out = (C.c_uint32 * 4)()
        self.L.emu_counters(self.e, out)
        return tuple(out)

    def seeds(self, strand):
        buf = (EmuSeed * 65536)(); ts = C.c_uint32()
        n = self.L.emu_seeds(self.e, strand, buf, 65536, C.byref(ts))
        return [(buf[i].offset, buf[i].len, buf[i].top) for i in range(n)], ts.value

    def scan_seeds(self, strand):
        """the seed list of one strand of the last read as k_seed_scan's per-lane state machine makes it (probing only what
        the scan consumes) -> (seeds, total_score, (probes with a valid k-mer, windows asked for))"""
        buf = (EmuSeed * 65536)(); ts = C.c_uint32(); pr = (C.c_uint32 * 2)()
        n = self.L.emu_scan_seeds(self.e, strand, buf, 65536, C.byref(ts), pr)
        return [(buf[i].offset, buf[i].len, buf[i].top) for i in range(n)], ts.value, (pr[0], pr[1])
