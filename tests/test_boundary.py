"""The C-ABI boundary without a GPU: the library loads, exports every symbol of include/desamba_amd.h,
reads the reference's index format, answers rank queries exactly like the reference's occ(), formats
SAM exactly like output_one_result_sam, and refuses to run without a gfx950 device."""
import ctypes as C
import os
import random
import re

import pytest

from conftest import GOLDEN, ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "desamba_amd.h")).read()
    return sorted(set(re.findall(r"\b(dsb_[a-z_0-9]+)\s*\(", txt)))


def test_exports_match_header(built):
    import desamba_amd as D
    L = D.lib()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), "libdesamba_amd.so does not export %s" % s
    assert sorted(D.EXPORTS) == syms


@pytest.mark.parametrize("rank64", [False, True])
def test_index_loader_and_rank_layout(demo, oracle, monkeypatch, rank64):
    """device rank layout (64-B blocks) == reference occ() (src/bwt.c:43-65) on random rows, every symbol; also with the
    64-bit superblock layout of indexes beyond 2^32 BWT symbols forced on (the demo BWT spans three superblocks)"""
    import desamba_amd as D
    if rank64:
        monkeypatch.setenv("DSB_FORCE_RANK64", "1")
    idx = D.Index(demo["index"])
    assert idx.n_ref == 463 and idx.ek_len == 16
    assert idx.ref_name(0).startswith("tid|")
    rng = random.Random(5)
    bwt_len = 11798750
    rows = [rng.randrange(bwt_len) for _ in range(20000)] + [0, 1, 127, 128, 129, 255, 256, bwt_len - 1] + \
        [k * (1 << 22) + d for k in (1, 2) for d in (-129, -128, -1, 0, 1, 127, 128)]
    for r in rows:
        for c in (0, 1, 2, 3, 4, 0xff):
            assert idx.occ_host(r, c) == oracle.occ(r, c)
    idx.close()


def test_compressed_hash_index_gives_the_same_intervals(demo):
    """the prefix table staged on the device as 64-byte lines of 29 prefixes (32-bit base + 16-bit offsets, 148 MB instead
    of 512 MiB) answers every lookup like hash_index[p], hash_index[p+1] of the on-disk table"""
    import ctypes as C
    import desamba_amd as D
    idx = D.Index(demo["index"]); L = D.lib()
    rng = random.Random(9)
    ps = [rng.randrange(1 << 26) for _ in range(200000)] + [0, 1, 28, 29, 30, 57, 58, (1 << 26) - 1, (1 << 26) - 29, (1 << 26) - 30]
    a0, a1, b0, b1 = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    nonempty = 0
    for p in ps:
        assert L.dsb_index_prefix_interval(idx.h, p, 0, C.byref(a0), C.byref(a1)) == 0
        assert L.dsb_index_prefix_interval(idx.h, p, 1, C.byref(b0), C.byref(b1)) == 0      # the demo index fits the compressed form
        assert (a0.value, a1.value) == (b0.value, b1.value), p
        nonempty += a1.value > a0.value
    assert nonempty > 1000
    idx.close()


def test_missing_index_is_an_error(built, tmp_path):
    import desamba_amd as D
    with pytest.raises(D.DsbError) as e:
        D.Index(str(tmp_path))
    assert e.value.code == D.DSB_EIO


def test_no_cpu_fallback(demo):
    """without a GPU the product path fails loudly (DSB_ENODEV); it never computes on the host"""
    import torch
    import desamba_amd as D
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    idx = D.Index(demo["index"])
    with pytest.raises(D.DsbError) as e:
        D.Ctx(idx, 0)
    assert e.value.code == D.DSB_ENODEV
    idx.close()


def test_sam_formatter_matches_reference_writer(demo, oracle, tmp_path):
    """dsb_format_sam over the oracle's hits reproduces the golden SAM bytes (incl. negative clips, H/S, MAPQ)"""
    import desamba_amd as D
    idx = D.Index(demo["index"])
    for name in ("pb", "ngs150", "ont5k_e25"):
        recs = D.read_fastq(os.path.join(GOLDEN, "synth", name + ".fq"))
        reads = D.make_reads(recs)
        out = []; hist = 0
        buf = C.create_string_buffer(1 << 16)
        for i, (nm, seq, q) in enumerate(recs):
            hits = oracle.classify(seq, hist); hist = max(hist, len(seq))
            arr = (D.DsbHit * max(len(hits), 1))()
            for k, h in enumerate(hits):
                arr[k].ref_ID, arr[k].t_st, arr[k].t_ed, arr[k].q_st, arr[k].q_ed, arr[k].sum_score, arr[k].direction, arr[k].primary, arr[k].pri_index = h
            n = D.lib().dsb_format_sam(idx.h, C.byref(reads[i]), arr, len(hits), 5, 0, buf, len(buf))
            assert n > 0
            out.append(buf.raw[:n])
        assert b"".join(out) == open(os.path.join(GOLDEN, "synth", name + ".ubfree.sam"), "rb").read()
    idx.close()


def test_cli_usage_without_args(built):
    import subprocess
    cli = os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA")
    p = subprocess.run([cli, "classify"], stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"Usage:" in p.stderr      # missing args -> usage, return 0 (src/cly_mt.c:504-508)
