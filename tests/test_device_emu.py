"""The DEVICE code (desamba_amd/csrc/dsb_classify_dev.h, dsb_probe.h) compiled for the host as a 1-lane
wave (tests/emu) against the oracle: seed-probe bits, seed lists, final hits.  Runs without a GPU."""
import os

import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def emu(demo):
    import emu_lib
    return emu_lib.Emu(demo["index"])


def _cmp(emu, oracle, recs, check_stages=False):
    """-> work-counter totals [device code, oracle] over the reads: occ, MEM searches, SA lookups, reference bases"""
    hist = 0
    tot = [[0] * 4, [0] * 4]
    for name, seq, q in recs:
        exp = oracle.classify(seq, hist)
        if check_stages:
            got, bF, bR = emu.classify(seq, hist, want_bits=True)
            if len(seq) >= 40:
                n = len(seq) - 16 + 1
                assert bytes(bF[:n]) == bytes(oracle.exist_bits(seq, 1)[:n])
                assert bytes(bR[:n]) == bytes(oracle.exist_bits(seq, 0)[:n])
                for s in (1, 0):
                    assert emu.seeds(s) == oracle.seeds(s)
        else:
            got = emu.classify(seq, hist)
        assert got == exp, name
        if len(seq) >= 40:
            # k_seed_scan's lane code (probes only what the scan consumes) gives the seed lists of the scan over all hit bits
            for s in (1, 0):
                sv, ts, _ = emu.scan_seeds(s)
                assert (sv, ts) == emu.seeds(s), (name, s)
        assert emu.n_anc() == oracle.n_anc(), name
        # the work counters behind the algorithmic bytes: occ, SA lookups, reference bases, MEM searches
        # (the device walks islands in parallel and commits in order, so an island that the reference skips after a
        # score > 512 -- src/cly.c:1530-1531, short high-identity reads -- is walked here and dropped at commit: its work counts)
        oc = oracle.counters(); ec = emu.counters(); t = (oc[2], oc[5], oc[3], oc[4])
        assert all(a >= b for a, b in zip(ec, t)), name
        for k in range(4):
            tot[0][k] += ec[k]; tot[1][k] += t[k]
        hist = max(hist, len(seq))
    return tot


def test_demo_reads(emu, oracle, demo):
    import desamba_amd as D
    _cmp(emu, oracle, D.read_fastq(demo["fastq"], 400), check_stages=True)


@pytest.mark.parametrize("name", ["ont20k", "ngs150", "pb", "ont5k_e25", "appc", "heavy", "wrapq", "ngs_e14", "overhang", "manyanchors"])
def test_synthetic(emu, oracle, name):
    import desamba_amd as D
    _cmp(emu, oracle, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")), check_stages=(name in ("ngs150", "appc")))


def test_edge_cases(emu, oracle):
    recs = [(b"short", b"ACGT" * 9, None), (b"min", b"ACGTTGCA" * 5, None), (b"polyA", b"A" * 300, None),
            (b"allN", b"N" * 200, None), (b"lower", b"acgtnnacgt" * 30, None), (b"l39", b"A" * 39, None), (b"empty", b"", None)]
    _cmp(emu, oracle, recs, check_stages=True)


def test_random_long_reads(emu, oracle, demo, tmp_path):
    """600 fresh 50 kbp ONT reads: wide enough to meet the rare paths (chains starting at q = -1, lane-scratch
    overflow of the island walk, repeats) that the small golden sets do not reach"""
    import subprocess
    from conftest import ROOT
    import desamba_amd as D
    fq = tmp_path / "ont50k.fq"
    subprocess.check_call([os.path.join(ROOT, "tools", "readsim"), demo["index"], str(fq), "600", "50000", "0.15", "1", "ont"])
    dev, ora = _cmp(emu, oracle, D.read_fastq(str(fq)))
    # the counters bench.py builds the algorithmic bytes from (dsb_timing.n_occ / n_mem / n_sa / ref_bases) agree with the
    # oracle's within 1 % on this workload
    for a, b in zip(dev, ora):
        assert b <= a <= 1.01 * b, (dev, ora)


def test_rank64_layout_in_device_code(oracle, demo, monkeypatch):
    """the device code with the 64-bit superblock rank layout forced on: same hits as the oracle"""
    import emu_lib
    import desamba_amd as D
    monkeypatch.setenv("DSB_FORCE_RANK64", "1")
    e = emu_lib.Emu(demo["index"])
    for name in ("ont20k", "wrapq", "appc"):
        _cmp(e, oracle, D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq")))


def test_second_index(strain, monkeypatch):
    """device code on the strain index: the heaviest reads of the set (right extensions across tandem repeats, up to
    67 k match nodes: beyond a slot's first-level arena, so the 1-lane harness gets the large one) and 16 others"""
    import emu_lib, oracle_lib, desamba_amd as D
    monkeypatch.setenv("DSB_EMU_SMS_CAP", "4000000")
    emu = emu_lib.Emu(strain["index"]); ora = oracle_lib.Oracle(strain["index"])
    recs = D.read_fastq(strain["fastq"])
    for i in [75, 23, 90] + list(range(16)):
        name, seq, q = recs[i]
        assert emu.classify(seq, 15000) == ora.classify(seq, 15000), name
        assert emu.n_anc() == ora.n_anc(), name
