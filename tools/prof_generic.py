"""Time an arbitrary FASTQ through the device path: python tools/prof_generic.py <reads.fq> [reps]
(index: $DSB_INDEX, default the demo index)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import desamba_amd as D
fq = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = D.Index(os.environ.get("DSB_INDEX", ROOT + "/data/demo/index")); ctx = D.Ctx(idx, 0)
t = time.time(); n = ctx.upload_fastq(fq); print("reads", n, "upload %.2fs" % (time.time() - t))
for _ in range(reps):
    t = time.time(); ctx.run(); dt = time.time() - t
tm = ctx.timing()
res = ctx.fetch(strict=False)
bad = sum(1 for i in range(n) if res.reads[i].status); mapped = sum(1 for i in range(n) if res.reads[i].n)
print("ms encode %.2f probe %.2f classify %.2f total %.2f | %.0f reads/s %.3f Gbp/s | mapped %d status!=0 %d second-run %d requeued-heavy %d tail %.1f" % (tm.encode_ms, tm.seed_probe_ms, tm.classify_ms, tm.total_ms, n / (tm.total_ms / 1e3), tm.bases / (tm.total_ms / 1e3) / 1e9, mapped, bad, tm.n_retry, tm.n_requeue, tm.tail_ms))
