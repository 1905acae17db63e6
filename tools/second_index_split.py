"""Stage split of k_classify on the bench's second index (321-Mbp synthetic strain collection, no short tandem repeats):
python3 tools/second_index_split.py [reads]   (sets DSB_DEBUG=1)"""
import os, subprocess, sys, shutil
os.environ["DSB_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = 50000
d = os.path.join(ROOT, "data", "bench_strain"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
fa = os.path.join(d, "syn.fa"); idxd = os.path.join(d, "index")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, "320", "11", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
st = D.build_index(fa, idxd); os.remove(fa)
idx = D.Index(idxd); gen = bench.Gen(idxd); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 777, 16)
ctx = D.Ctx(idx, 0, max_read_len=L, max_batch_reads=n)
ctx.upload_text(p, nb, off, ln, n)
ctx.run(); ctx.run()
t = ctx.timing(); res = ctx.fetch(strict=False)
us = sorted((res.reads[i].device_us for i in range(n)), reverse=True)
print("ms: seed %.1f classify %.1f total %.1f | wave time mean %.1f ms median %.1f p99 %.1f max %.1f | anchors mean %.0f" % (t.seed_probe_ms, t.classify_ms, t.total_ms,
      sum(us) / n / 1e3, us[n // 2] / 1e3, us[n // 100] / 1e3, us[0] / 1e3, sum(res.reads[i].n_anc for i in range(n)) / n))
shutil.rmtree(d, ignore_errors=True)
