"""a-9 `lv_extd` (src/cly.c:510-609) on its own (it was covered only through whole-read SAM identity).

The device code's register-packed form (dsb_classify_dev.h: both strings as 3-bit symbols of one 64-bit word, the two tables of the
recurrence as 4-bit fields) compiled for the host (tests/emu) against the oracle's restatement of the reference's byte loops, on the
inputs its callers can produce: two strings of one length 0..12, the 8 bytes in front of the reference string never match, the bytes in
front of the query are either never-matching (a local copy) or real bases (a string inside the read buffer)."""
import ctypes as C
import os
import random

import emu_lib
import oracle_lib

PAD_Q, PAD_T = 0xF1, 0xF2          # oracle/classify.c LVPAD_Q / LVPAD_T


def _libs():
    O = oracle_lib.lib()
    E = C.CDLL(emu_lib.EMU_SO)
    for L, f in ((O, "ora_lv_extd"), (E, "emu_lv_extd")):
        getattr(L, f).argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]; getattr(L, f).restype = C.c_int32
    return O, E


def _cases(rng, n):
    for _ in range(n):
        ln = rng.randint(0, 12)
        ref = [rng.randint(0, 3) for _ in range(ln)]
        kind = rng.random()
        if kind < 0.15:
            qry = list(ref)
        elif kind < 0.3:
            qry = [rng.randint(0, 3) for _ in range(ln)]
        else:                                   # the reference string with a few substitutions, insertions, deletions: what map_seed extends over
            qry = []
            for b in ref:
                r = rng.random()
                if r < 0.12:
                    qry.append(rng.randint(0, 3))
                elif r < 0.2:
                    continue
                elif r < 0.28:
                    qry += [b, rng.randint(0, 3)]
                else:
                    qry.append(b)
            qry = (qry + [rng.randint(0, 3) for _ in range(12)])[:ln]
        low = rng.choice([1, 2]) if rng.random() < 0.1 else 4          # low-complexity strings: many equal diagonals
        if low < 4:
            ref = [b % low for b in ref]; qry = [b % low for b in qry]
        front_q = bytes([PAD_Q] * 8) if rng.random() < 0.5 else bytes(rng.randint(0, 3) for _ in range(8))
        yield bytes([PAD_T] * 8) + bytes(ref), front_q + bytes(qry), ln


def test_device_lv_extd_equals_the_reference_recurrence():
    O, E = _libs()
    rng = random.Random(9)
    n = bad = 0
    dist = {}
    for ref, qry, ln in _cases(rng, 60000):
        a = O.ora_lv_extd(ref, ln, qry, ln); b = E.emu_lv_extd(ref, ln, qry, ln)
        dist[a] = dist.get(a, 0) + 1
        n += 1
        if a != b:
            bad += 1
            assert bad < 5, "lv_extd differs: ref %r query %r len %d: oracle %d device %d" % (ref[8:], qry, ln, a, b)
    assert bad == 0
    assert n == 60000 and len(dist) >= 6, dist          # distances 0 .. 4 and the "more than four" results all occur
