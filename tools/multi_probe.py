"""Where the time of dsb_multi_classify_batch goes on one device listed twice, against one context on the whole batch:
python3 tools/multi_probe.py [n_reads] [len]   (demo index; prints wall times and the device timings of every context)"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import __graft_entry__ as G
import desamba_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072; L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
d = G.demo_dir(); idxd = os.path.join(d, "index")
fq = "/dev/shm/dsb_multi_probe.fq"
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fastq.py"), idxd, fq, str(n), str(L), "0.15", "77", "ont", "8"], check=True, stdout=subprocess.DEVNULL)
lines = open(fq, "rb").read().split(b"\n"); os.remove(fq)
recs = [(lines[i][1:], lines[i + 1], None) for i in range(0, len(lines) - 3, 4)]; del lines
reads = D.make_reads(recs)
idx = D.Index(idxd)
ctx = D.Ctx(idx, 0, max_read_len=L + 64, max_batch_reads=n, max_batch_bases=n * (L + 100))
ctx.classify(reads)
for rep in range(2):
    t0 = time.perf_counter(); ctx.upload(reads); t1 = time.perf_counter(); ctx.run(); t2 = time.perf_counter(); ctx.fetch(); t3 = time.perf_counter()
    t = ctx.timing()
    print("single: upload %.3f run %.3f fetch %.3f s; device total %.1f ms (encode %.1f order %.1f seed %.1f classify %.1f tail %.1f), upload %.0f MB" %
          (t1 - t0, t2 - t1, t3 - t2, t.total_ms, t.encode_ms, t.order_ms, t.seed_probe_ms, t.classify_ms, t.tail_ms, t.upload_bytes / 1e6), flush=True)
ctx.close()
m = D.Multi(idx, [0, 0])
m.classify(reads)
for rep in range(2):
    m.reset_history()
    t0 = time.perf_counter(); m.classify(reads); t1 = time.perf_counter()
    print("multi [0, 0]: %.3f s, calls %r" % (t1 - t0, m.last_calls()), flush=True)
    for i in range(2):
        t = D.DsbTiming(); D.lib().dsb_batch_timing(D.lib().dsb_multi_ctx(m.h, i), C.byref(t))
        print("   ctx %d: device total %.1f ms (encode %.1f order %.1f seed %.1f classify %.1f tail %.1f)" % (i, t.total_ms, t.encode_ms, t.order_ms, t.seed_probe_ms, t.classify_ms, t.tail_ms), flush=True)
m.close(); idx.close()
