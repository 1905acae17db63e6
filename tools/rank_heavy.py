"""Where do the slowest reads of the bench batch rank by the repeat score of k_repeat_score, and by a tandem-specific score?"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = 65536; L = 50000
import __graft_entry__ as G
idx_dir = os.path.join(G.demo_dir(), "index")       # built from the committed demo zips when it is not there
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)
ctx = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=n)
ctx.upload_text(p, nb, off, ln, n); ctx.run(); ctx.run()
res = ctx.fetch(strict=False)
us = np.array([res.reads[i].device_us for i in range(n)])
lut = np.zeros(256, dtype=np.uint64); lut[ord('C')] = 1; lut[ord('G')] = 2; lut[ord('T')] = 3
raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb,))
def scores(i):
    c = lut[raw[off[i]:off[i] + ln[i]]]
    k = np.zeros(len(c) - 11, dtype=np.uint64)
    for j in range(12):
        k = (k << np.uint64(2)) | c[j:len(c) - 11 + j]
    k2 = k[::2]
    rep = len(k2) - len(np.unique(k2))
    # tandem-specific: 12-mers equal to the 12-mer d positions earlier, for small d
    t = 0
    for dd in (2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64):
        t = max(t, int((k[dd:] == k[:-dd]).sum()))
    return rep, t
# ranks are estimated against a random sample of the batch (scoring all 65536 reads in numpy takes minutes)
rng = np.random.default_rng(1); sample = rng.choice(n, 3000, replace=False)
ss = np.array([scores(int(i)) for i in sample])
top = np.argsort(-us)[:32]
print("kernel ms: classify %.1f, tail %.1f; wave time of the batch: mean %.2f ms, p99 %.1f, p99.9 %.1f" % (ctx.timing().classify_ms, ctx.timing().tail_ms, us.mean() / 1e3, np.percentile(us, 99) / 1e3, np.percentile(us, 99.9) / 1e3))
print("slowest reads: (ms, estimated rank by repeat score, by tandem score, rep, tand)")
for i in top:
    r, t = scores(int(i))
    print("  %.1f  %6d  %6d  %6d %6d" % (us[i] / 1e3, int((ss[:, 0] > r).mean() * n), int((ss[:, 1] > t).mean() * n), r, t))
# the stage split of the slowest reads alone (needs a -DDSB_TIMERS build for the fine timers): DSB_RANK_HEAVY_SPLIT=1
if os.environ.get("DSB_RANK_HEAVY_SPLIT"):
    sel = [int(i) for i in top]
    off2 = (C.c_uint64 * len(sel))(*[off[i] for i in sel]); ln2 = (C.c_uint32 * len(sel))(*[ln[i] for i in sel])
    os.environ["DSB_DEBUG"] = "1"
    ctx2 = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=len(sel))
    ctx2.upload_text(p, nb, off2, ln2, len(sel)); ctx2.run()
    r2 = ctx2.fetch(strict=False)
    print("alone: %s ms" % " ".join("%.0f" % (r2.reads[i].device_us / 1e3) for i in range(len(sel))))
