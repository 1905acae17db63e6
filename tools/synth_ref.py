#!/usr/bin/env python3
"""Deterministic synthetic reference collection for index-size experiments:
`synth_ref.py <out.fa> <Mbp> [seed [tandem_hi unit_hi [unit_lo]]]` writes about <Mbp> million bases of FASTA: random base genomes of
60-400 kbp, each followed by 0-3 strains that differ from it by 0.5-4 % substitutions and a few short
indels (so the de Bruijn graph branches the way a RefSeq collection does), a pool of 1-4 kbp mobile elements
copied into random genomes, and short tandem repeats (0..tandem_hi-1 per genome, default 3, unit length
unit_lo..unit_hi-1, default 2..59).  Headers follow the reference's
`>tid|<n>|ref|<name>` convention."""
import sys
import numpy as np


def main():
    out, mbp = sys.argv[1], float(sys.argv[2])
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    tandem_hi = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    unit_hi = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    unit_lo = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    mobile = [rng.integers(0, 4, rng.integers(1000, 4000), dtype=np.uint8) for _ in range(64)]
    total, target, gid = 0, int(mbp * 1e6), 0
    with open(out, "wb") as f:
        def emit(seq, tag):
            nonlocal total, gid
            gid += 1
            f.write(b">tid|%d|ref|SYN_%06d.%s synthetic\n" % (100000 + gid, gid, tag.encode()))
            txt = acgt[seq]
            for i in range(0, len(txt), 70 * 4096):
                blk = txt[i:i + 70 * 4096]
                rows = [blk[j:j + 70].tobytes() for j in range(0, len(blk), 70)]
                f.write(b"\n".join(rows) + b"\n")
            total += len(seq)

        while total < target:
            n = int(rng.integers(60000, 400000))
            g = rng.integers(0, 4, n, dtype=np.uint8)
            for _ in range(int(rng.integers(0, 4))):             # mobile elements
                m = mobile[int(rng.integers(0, len(mobile)))]
                p = int(rng.integers(0, n - len(m)))
                g[p:p + len(m)] = m
            for _ in range(int(rng.integers(0, tandem_hi))):     # tandem repeats
                unit = rng.integers(0, 4, int(rng.integers(unit_lo, unit_hi)), dtype=np.uint8)
                ln = int(rng.integers(200, 3000))
                p = int(rng.integers(0, n - ln))
                g[p:p + ln] = np.resize(unit, ln)
            emit(g, "1")
            for s in range(int(rng.integers(0, 4))):             # strains
                div = rng.uniform(0.005, 0.04)
                h = g.copy()
                mut = rng.random(n) < div
                h[mut] = (h[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
                cuts = np.sort(rng.integers(0, n, int(rng.integers(2, 20))))
                parts, prev = [], 0
                for c in cuts:                                   # short indels at the cut points
                    parts.append(h[prev:c])
                    if rng.random() < 0.5:
                        parts.append(rng.integers(0, 4, int(rng.integers(1, 40)), dtype=np.uint8))
                        prev = c
                    else:
                        prev = min(n, c + int(rng.integers(1, 40)))
                parts.append(h[prev:])
                emit(np.concatenate(parts), "s%d" % (s + 2))
    sys.stderr.write("synth_ref: %d sequences, %d bases\n" % (gid, total))


if __name__ == "__main__":
    main()
