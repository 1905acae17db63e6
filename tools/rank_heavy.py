"""Where do the slowest reads of the bench batch rank by the repeat score of k_repeat_score, and by a tandem-specific score?"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = 65536; L = 50000
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)
ctx = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=n)
ctx.upload_text(p, nb, off, ln, n); ctx.run(); ctx.run()
res = ctx.fetch(strict=False)
us = np.array([res.reads[i].device_us for i in range(n)])
lut = np.zeros(256, dtype=np.uint64); lut[ord('C')] = 1; lut[ord('G')] = 2; lut[ord('T')] = 3
rep = np.zeros(n); tand = np.zeros(n)
raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb,))
for i in range(n):
    c = lut[raw[off[i]:off[i] + ln[i]]]
    k = np.zeros(len(c) - 11, dtype=np.uint64)
    for j in range(12):
        k = (k << np.uint64(2)) | c[j:len(c) - 11 + j]
    k2 = k[::2]
    rep[i] = len(k2) - len(np.unique(k2))
    # tandem-specific: 12-mers equal to the 12-mer d positions earlier, for small d
    t = 0
    for dd in (2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64):
        t = max(t, int((k[dd:] == k[:-dd]).sum()))
    tand[i] = t
order_rep = np.argsort(-rep); rank_rep = np.empty(n, int); rank_rep[order_rep] = np.arange(n)
order_t = np.argsort(-tand); rank_t = np.empty(n, int); rank_t[order_t] = np.arange(n)
top = np.argsort(-us)[:24]
print("slowest reads: (ms, rank by repeat score, rank by tandem score, rep, tand)")
for i in top: print("  %.1f  %6d  %6d  %6d %6d" % (us[i] / 1e3, rank_rep[i], rank_t[i], rep[i], tand[i]))
print("corr rep %.3f tand %.3f" % (np.corrcoef(rep, us)[0, 1], np.corrcoef(tand, us)[0, 1]))
