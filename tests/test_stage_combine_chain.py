"""a-11 `sc_hash_idx` / `combine_chain` (src/cly.c:1691-1710,1763-1808) on their own.

Random chain tables (several chains per diagonal byte so that the 256 buckets chain up, chains that end where others begin, both
strands, several references) are indexed and then asked, in order, the questions the extensions ask -- with the questions aimed at
existing chain ends most of the time so that merges happen and later questions see their effect (merged-away chains have no score
left).  The device code compiled for the host must give the oracle's answers and leave the same table; its read-only `combine_test`
(what the block-wise extension asks for 64 nodes at once) must predict every answer."""
import ctypes as C
import random

import emu_lib
import oracle_lib


def test_device_combine_chain_equals_the_reference():
    O = oracle_lib.lib(); E = C.CDLL(emu_lib.EMU_SO)
    O.ora_combine_stage.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]; O.ora_combine_stage.restype = None
    E.emu_combine_stage.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]; E.emu_combine_stage.restype = None
    rng = random.Random(4)
    merges = 0
    for rnd in range(400):
        n = rng.randint(1, 60)
        rows = []
        for i in range(n):
            q_st = rng.randint(0, 30000); ln = rng.randint(20, 3000)
            t_st = rng.randint(0, 200000) if rng.random() < 0.4 or not rows else rows[-1][6] + rng.randint(-8, 60)    # often right behind the previous chain
            t_st = max(t_st, 0)
            if rows and rng.random() < 0.6:
                q_st = rows[-1][8] + (t_st - rows[-1][6]) + rng.choice([0, 0, 0, 1, -1, 256, -256])        # same diagonal (or one that shares its bucket)
                q_st = max(q_st, 0)
            rows.append([rng.randint(0, 2), rng.randint(0, 1), rng.choice([0, 0, 50, 300, 4000]) if rng.random() < 0.2 else rng.randint(64, 5000), rng.randint(1, 40), rng.randint(0, 30),
                         t_st, t_st + ln + rng.randint(-10, 10), q_st, q_st + ln])
        qs = []
        for _ in range(rng.randint(1, 120)):
            cid = rng.randint(0, n - 1); isleft = rng.randint(0, 1)
            if rng.random() < 0.75:                                                 # aim at an end of some other chain
                o = rows[rng.randint(0, n - 1)]
                dis = (o[6] - o[8]) if isleft else (o[5] - o[7])
                qp = ((o[8] - 9) if isleft else o[7]) + rng.choice([0, 0, 3, -3, 7, -7, 8, -8, 20])
            else:
                dis = rng.randint(-70000, 230000); qp = rng.randint(0, 33000)
            qs.append([cid, dis, isleft, qp])
        A = (C.c_uint32 * (9 * n))(*[v & 0xffffffff for r in rows for v in r]); B = (C.c_uint32 * (9 * n))(*A)
        Q = (C.c_int32 * (4 * len(qs)))(*[v for q in qs for v in q])
        oa = (C.c_int32 * len(qs))(); ob = (C.c_int32 * len(qs))(); tb = (C.c_int32 * len(qs))()
        O.ora_combine_stage(A, n, Q, len(qs), oa)
        E.emu_combine_stage(B, n, Q, len(qs), ob, tb)
        assert list(oa) == list(ob), (rnd, list(oa), list(ob))
        assert list(A) == list(B), rnd
        assert [1 if v >= 0 else 0 for v in ob] == list(tb), rnd
        merges += sum(1 for v in oa if v >= 0)
    assert merges > 300, merges
