"""What is a perfect longest-first order worth?  One bench-like batch is run with the library's own order, then again with
the reads ordered by the wave times measured in the first run (DSB_ORDER_FILE): python3 tools/oracle_order.py [reads]"""
import os, sys, struct
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
def run(tag):
    ctx.run(); ctx.run(); t = ctx.timing()
    print("%-28s seed %.1f classify %.1f tail %.1f total %.1f ms" % (tag, t.seed_probe_ms, t.classify_ms, t.tail_ms, t.total_ms))
    return ctx.fetch(strict=False)
res = run("library order:")
us = [res.reads[i].device_us for i in range(n)]
order = sorted(range(n), key=lambda i: -us[i])
path = "/tmp/dsb_order.bin"
open(path, "wb").write(struct.pack("<%dI" % n, *order))
os.environ["DSB_ORDER_FILE"] = path
run("measured wave times as order:")
tot = sum(us) / 1e3
print("sum of wave times %.0f ms / 3072 slots = %.1f ms" % (tot, tot / 3072))
