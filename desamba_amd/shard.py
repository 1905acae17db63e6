"""Read sharding for N GPUs of one node: contiguous chunks of the input order dealt
round-robin over ranks, index replicated, no collective on the data path.  The only cross-read state of
the reference -- the running max_read_l of delete_small_score_rst (src/cly.c:2958) -- travels in the
chunk header as the prefix maximum of read length before the chunk (oracle U4)."""


def plan(lengths, world, chunk_bases=64_000_000, chunk_reads=4096):
    """-> list over ranks of [(start, end, hist_max_before), ...] covering range(len(lengths)) exactly once"""
    chunks, start, bases, hist = [], 0, 0, 0
    run_max = 0
    for i, L in enumerate(lengths):
        bases += L
        run_max = max(run_max, L)
        if bases >= chunk_bases or i + 1 - start >= chunk_reads or i + 1 == len(lengths):
            chunks.append((start, i + 1, hist))
            hist = max(hist, run_max)
            start, bases = i + 1, 0
    out = [[] for _ in range(world)]
    for k, c in enumerate(chunks):
        out[k % world].append(c)
    return out


def merge(per_rank_results, plans):
    """per_rank_results[r] = list (per chunk of plans[r]) of per-read result lists -> one list in input order"""
    n = max((c[1] for p in plans for c in p), default=0)
    merged = [None] * n
    for r, p in enumerate(plans):
        for k, (s, e, _) in enumerate(p):
            merged[s:e] = per_rank_results[r][k]
    return merged
