"""Slowest reads of a bench-like batch: python tools/tail_reads.py [reads]  (env DSB_HEAVY_MW etc. apply)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = 50000
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)
ctx = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=n)
ctx.upload_text(p, nb, off, ln, n)
ctx.run(); ctx.run()
t = ctx.timing()
res = ctx.fetch(strict=False)
us = sorted(((res.reads[i].device_us, i) for i in range(n)), reverse=True)
print("ms: seed %.1f classify %.1f tail %.1f total %.1f  early %d mw %d" % (t.seed_probe_ms, t.classify_ms, t.tail_ms, t.total_ms, t.n_early, t.n_heavy_mw))
print("slowest reads (ms, index):", [(round(u / 1e3, 1), i) for u, i in us[:12]])
