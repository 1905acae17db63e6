"""N>1 path on CPU: two gloo ranks shard a mixed-length input (desamba_amd.shard), each classifies its
chunks with the host emulation of the device code, rank 0 gathers and must equal the unsharded oracle run
(including the max_read_l carry that decides the 2G / 3G filter mode for short reads)."""
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _worker(rank, world, port, index_dir, fq_paths, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import desamba_amd as D
    from desamba_amd import shard
    import emu_lib
    recs = []
    for p in fq_paths:
        recs += D.read_fastq(p)
    plans = shard.plan([len(r[1]) for r in recs], world, chunk_bases=60000, chunk_reads=50)
    emu = emu_lib.Emu(index_dir)
    mine = []
    for (s, e, hist) in plans[rank]:
        out = []
        for (nm, seq, ql) in recs[s:e]:
            out.append(emu.classify(seq, hist)); hist = max(hist, len(seq))
        mine.append(out)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank == 0:
        q.put(shard.merge(gathered, plans))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharding_equals_single_run(demo, oracle):
    import desamba_amd as D
    fqs = [os.path.join(GOLDEN, "synth", "ngs150.fq"), os.path.join(GOLDEN, "synth", "pb.fq"), os.path.join(GOLDEN, "synth", "appc.fq")]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, demo["index"], fqs, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    recs = []
    for p in fqs:
        recs += D.read_fastq(p)
    hist = 0
    assert len(merged) == len(recs)
    for (nm, seq, ql), got in zip(recs, merged):
        assert got == oracle.classify(seq, hist), nm
        hist = max(hist, len(seq))


def test_plan_covers_everything_once():
    from desamba_amd import shard
    lens = [150] * 1000 + [50000] * 7 + [3000] * 40
    for world in (1, 2, 3, 8):
        plans = shard.plan(lens, world, chunk_bases=100000, chunk_reads=64)
        seen = sorted((s, e) for p in plans for (s, e, h) in p)
        assert seen[0][0] == 0 and seen[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
        for p in plans:
            for (s, e, h) in p:
                assert h == max(lens[:s], default=0)
