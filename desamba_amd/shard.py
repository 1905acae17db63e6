"""Read sharding for N GPUs of one node: contiguous chunks of the input order dealt
round-robin over ranks, index replicated, no collective on the data path.  The only cross-read state of
the reference -- the running max_read_l of delete_small_score_rst (src/cly.c:2958) -- travels in the
chunk header as the prefix maximum of read length before the chunk (oracle U4).

The rule itself is dsb_shard_plan in libdesamba_amd.so (the one dsb_multi_classify_batch and a multi-process
host use); this module is its Python face for tests and bench.py."""
from . import shard_plan


def plan(lengths, world, chunk_bases=64_000_000, chunk_reads=4096):
    """-> list over ranks of [(start, end, hist_max_before), ...] covering range(len(lengths)) exactly once"""
    out = [[] for _ in range(world)]
    for (s, e, hist, rank) in shard_plan(lengths, world, chunk_bases, chunk_reads):
        out[rank].append((s, e, hist))
    return out


def merge(per_rank_results, plans):
    """per_rank_results[r] = list (per chunk of plans[r]) of per-read result lists -> one list in input order"""
    n = max((c[1] for p in plans for c in p), default=0)
    merged = [None] * n
    for r, p in enumerate(plans):
        for k, (s, e, _) in enumerate(p):
            merged[s:e] = per_rank_results[r][k]
    return merged
