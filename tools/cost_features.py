"""How well do cheap per-read features predict a read's wave time?  python3 tools/cost_features.py [reads]
Features from the seed lists (count, top count, total length per strand) against res.reads[i].device_us of a bench-like batch."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = 50000
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)
recs = [(b"r%d" % i, C.string_at(p + off[i], ln[i]), None) for i in range(n)]
reads = D.make_reads(recs)
ctx = D.Ctx(idx, 0)
ctx.upload(reads); ctx.run(); ctx.run()
res = ctx.fetch(strict=False)
us = np.array([res.reads[i].device_us for i in range(n)], dtype=np.float64)
F = []
for i in range(n):
    f = []
    for s in (0, 1):
        sv, ts = ctx.seeds(i, s)
        f += [len(sv), sum(1 for x in sv if x[2]), sum(x[1] for x in sv), ts]
    F.append(f)
F = np.array(F, dtype=np.float64)
# the repeat score of k_repeat_score, exactly: 12-mers at even positions that occurred before in the read
lut = np.zeros(256, dtype=np.uint64); lut[ord('C')] = 1; lut[ord('G')] = 2; lut[ord('T')] = 3
rep = np.zeros(n)
for i in range(n):
    c = lut[np.frombuffer(recs[i][1], dtype=np.uint8)]
    k = np.zeros(len(c) - 11, dtype=np.uint64)
    for j in range(12):
        k = (k << np.uint64(2)) | c[j:len(c) - 11 + j]
    k = k[::2]
    rep[i] = len(k) - len(np.unique(k))
print("repeat score corr %.3f   log2 bucket corr %.3f" % (np.corrcoef(rep, us)[0, 1], np.corrcoef(np.floor(np.log2(rep + 1)), us)[0, 1]))
top1 = set(np.argsort(-us)[: n // 100])
print("heaviest 1 %% in the top 5 %% by repeat score: %d of %d; by nF+nR: %d; by repeat bucket then nF+nR: %d" % (
    len(top1 & set(np.argsort(-rep)[: n // 20])), len(top1), len(top1 & set(np.argsort(-(F[:, 0] + F[:, 4]))[: n // 20])),
    len(top1 & set(np.lexsort((-(F[:, 0] + F[:, 4]), -np.floor(np.log2(rep + 1))))[: n // 20]))))
A2 = np.column_stack([F[:, 0] + F[:, 4], rep, np.ones(n)]); c2, *_ = np.linalg.lstsq(A2, us, rcond=None); p2 = A2 @ c2
print("fit on (nF+nR, repeat): R^2 %.3f coef %s; heaviest 1 %% in its top 5 %%: %d" % (1 - ((us - p2) ** 2).sum() / ((us - us.mean()) ** 2).sum(), np.round(c2, 2), len(top1 & set(np.argsort(-p2)[: n // 20]))))
names = ["nF", "topF", "lenF", "totF", "nR", "topR", "lenR", "totR"]
for k, nm in enumerate(names):
    print("%-5s corr %.3f" % (nm, np.corrcoef(F[:, k], us)[0, 1]))
both = F[:, 0] + F[:, 4]; print("nF+nR corr %.3f" % np.corrcoef(both, us)[0, 1])
mx = np.maximum(F[:, 3], F[:, 7]); mn = np.minimum(F[:, 3], F[:, 7])
print("max(tot) corr %.3f  min(tot) corr %.3f  anchors corr %.3f" % (np.corrcoef(mx, us)[0, 1], np.corrcoef(mn, us)[0, 1], np.corrcoef(np.array([res.reads[i].n_anc for i in range(n)], dtype=np.float64), us)[0, 1]))
A = np.column_stack([F, np.ones(n)]); coef, *_ = np.linalg.lstsq(A, us, rcond=None)
pred = A @ coef; print("linear fit R^2 %.3f" % (1 - ((us - pred) ** 2).sum() / ((us - us.mean()) ** 2).sum()), "coef", np.round(coef, 2))
print("us mean %.0f p50 %.0f p99 %.0f max %.0f" % (us.mean(), np.median(us), np.percentile(us, 99), us.max()))
# the heaviest 1 %: how many of them are in the top 5 % by the fit?
top = set(np.argsort(-us)[: n // 100]); top_pred = set(np.argsort(-pred)[: n // 20])
print("heaviest 1 %% found in the predicted top 5 %%: %d of %d" % (len(top & top_pred), len(top)))
