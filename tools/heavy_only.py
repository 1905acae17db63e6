"""The slowest reads of the bench batch on their own (no other load): per-read wave time with and without the multi-wave kernel"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = 65536; L = 50000
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)
sel = [21607, 45890, 32109, 37677, 37689, 22465, 54604, 22040, 32520, 45158, 49699, 47294, 4571, 207, 52165, 47570]
recs = [(b"r%d" % i, C.string_at(p + off[i], ln[i]), None) for i in sel] + [(b"f%d" % i, C.string_at(p + off[i], ln[i]), None) for i in range(16)]
reads = D.make_reads(recs)
for mw in ("0", "16"):
    os.environ["DSB_HEAVY_FIRST"] = "16"; os.environ["DSB_HEAVY_MW"] = mw
    ctx = D.Ctx(idx, 0)
    ctx.upload(reads); ctx.run(); ctx.run()
    res = ctx.fetch(strict=False); t = ctx.timing()
    print("MW=%s total %.1f ms (mw reads %d):" % (mw, t.total_ms, t.n_heavy_mw), [round(res.reads[i].device_us / 1e3, 1) for i in range(len(sel))])
    ctx.close()
