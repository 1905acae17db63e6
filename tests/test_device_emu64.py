"""The DEVICE code (desamba_amd/csrc/dsb_classify_dev.h) as 64 lanes with a race detector (tests/emu/emu_simt.cpp: a bulk-synchronous
machine -- a lane runs from one cross-lane operation to the next on the memory of the stretch's start, the lanes' stores are put in
place together) against the oracle: every cross-lane operation, every lane's indexing, the one-node-per-lane sparse DP of the
extensions, one island per lane, one gap per lane.  No finding is allowed: no two lanes leaving different values in one place, no lane
reading what another changes in the same stretch, no access outside the arrays, no cross-lane operation in divergent control flow.
Runs without a GPU."""
import ctypes as C
import os

import pytest

from conftest import GOLDEN, ROOT

LIB64 = os.path.join(ROOT, "tests", "emu", "libdsbemu64.so")


@pytest.fixture(scope="module")
def emu64(demo, built):
    import importlib
    import emu_lib
    old = os.environ.get("DSB_EMU_LIB")
    os.environ["DSB_EMU_LIB"] = LIB64
    try:
        importlib.reload(emu_lib)
        e = emu_lib.Emu(demo["index"])
    finally:
        if old is None:
            os.environ.pop("DSB_EMU_LIB", None)
        else:
            os.environ["DSB_EMU_LIB"] = old
        importlib.reload(emu_lib)
    return e


def run(emu64, oracle, recs):
    hist = 0
    for name, seq, q in recs:
        exp = oracle.classify(seq, hist)
        got = emu64.classify(seq, hist)
        assert got == exp, name
        f = emu64.findings()
        assert not f, (name, f)
        assert emu64.n_anc() == oracle.n_anc(), name
        hist = max(hist, len(seq))


def test_detector_sees_what_it_should(emu64):
    """four tiny kernels: a clean exchange across a wave_sync, conflicting stores, a read of what the neighbour writes in the same
    stretch, a store behind the array"""
    buf = C.create_string_buffer(4096)
    emu64.L.emu_selftest.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    assert emu64.L.emu_selftest(0, buf, len(buf)) == 0
    assert emu64.L.emu_selftest(1, buf, len(buf)) >= 1 and b"conflict" in buf.value
    assert emu64.L.emu_selftest(2, buf, len(buf)) >= 1 and b"race" in buf.value
    assert emu64.L.emu_selftest(3, buf, len(buf)) >= 1 and b"bounds" in buf.value


def test_demo_reads(emu64, oracle, demo):
    import desamba_amd as D
    run(emu64, oracle, D.read_fastq(demo["fastq"], 250))


@pytest.mark.parametrize("name,limit", [("ont20k", 12), ("pb", 12), ("ngs150", 150), ("appc", None), ("wrapq", None), ("overhang", None), ("ont5k_e25", 20), ("ngs_e14", 80), ("manyanchors", 1)])
def test_synthetic(emu64, oracle, name, limit):
    import desamba_amd as D
    run(emu64, oracle, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq"), limit))


def test_edge_cases(emu64, oracle):
    recs = [(b"short", b"ACGT" * 9, None), (b"min", b"ACGTTGCA" * 5, None), (b"polyA", b"A" * 300, None),
            (b"allN", b"N" * 200, None), (b"lower", b"acgtnnacgt" * 30, None), (b"l39", b"A" * 39, None), (b"empty", b"", None)]
    run(emu64, oracle, recs)


def test_lanes_in_reverse_order(emu64, oracle, demo, monkeypatch):
    """the lanes take their turns from 63 down (LDS atomics hand out places in another order): same hits"""
    import desamba_amd as D
    monkeypatch.setenv("DSB_EMU_ORDER", "rev")
    run(emu64, oracle, D.read_fastq(os.path.join(GOLDEN, "synth", "ont20k.fq"), 4) + D.read_fastq(demo["fastq"], 60))
