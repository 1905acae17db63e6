"""1 M x 150 bp reads on the demo index (bench.py's config2_short_reads) on their own: python3 tools/short_reads.py  (env knobs apply)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = 1 << 20; L = 150
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 48) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.01, 4242, 16)
ctx = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=n)
ctx.upload_text(p, nb, off, ln, n)
ms = []
for _ in range(4):
    ctx.run(); t = ctx.timing(); ms.append(t.total_ms)
print("ms", [round(m, 1) for m in ms], "-> %.2f M reads/s (classify %.1f ms)" % (n / sorted(ms[1:])[1] / 1e3, t.classify_ms))
