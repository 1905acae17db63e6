"""Per-stage wave-time split of k_classify (DSB_DEBUG=1): python tools/stage_split.py [reads] [read_len]"""
import os, sys, ctypes as C
os.environ["DSB_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)
ctx = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=n)
ctx.upload_text(p, nb, off, ln, n)
ctx.run(); ctx.run()
t = ctx.timing()
print("ms: encode %.2f order %.2f seed %.2f classify %.2f tail %.2f (seed_scan=%d)" % (t.encode_ms, t.order_ms, t.seed_probe_ms, t.classify_ms, t.tail_ms, t.seed_scan))
