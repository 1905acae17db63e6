"""Small fixed workload for rocprofv3: N synthetic 50 kbp reads through the three kernels, `reps` times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import desamba_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fq = "/tmp/prof_%d.fq" % n
if not os.path.exists(fq):
    os.system("%s/tools/readsim %s/data/demo/index %s %d 50000 0.15 1 ont" % (ROOT, ROOT, fq, n))
idx = D.Index(ROOT + "/data/demo/index"); ctx = D.Ctx(idx, 0, n_slots=slots)
recs = D.read_fastq(fq)
if os.environ.get("ONLY"):
    recs = [recs[int(i)] for i in os.environ["ONLY"].split(",")]
reads = D.make_reads(recs); ctx.upload(reads)
for _ in range(reps):
    ctx.run()
t = ctx.timing()
print("ms encode %.3f probe %.3f classify %.3f" % (t.encode_ms, t.seed_probe_ms, t.classify_ms))
res = ctx.fetch(strict=False)
us = sorted((res.reads[i].device_us, i, res.reads[i].n, res.reads[i].status) for i in range(len(reads)))
import statistics
print("per-read device us: min %d median %d p90 %d p99 %d max %d sum %.1f ms" % (us[0][0], us[len(us)//2][0], us[int(len(us)*0.9)][0], us[int(len(us)*0.99)][0], us[-1][0], sum(u[0] for u in us)/1e3))
print("slowest:", us[-8:])
