"""Only the seed lookup on synthetic multi-GiB filter tables (bench.py's roofline_seed_lookup_hbm hook), for a PMC pass
of its own: under rocprofv3 the k_seed_scan launches of this program are all in the HBM regime.
   python3 tools/seed_hbm_only.py [reads] [table MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, desamba_amd as D
import __graft_entry__ as G
G.demo_dir()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
L = 50000
idx_dir = os.path.join(ROOT, "data", "demo", "index")
idx = D.Index(idx_dir); gen = bench.Gen(idx_dir); lib = D.lib()
cap = n * (2 * L + 40) + (1 << 20)
p = lib.dsb_host_alloc(cap)
nb, off, ln = gen.fill(p, cap, n, L, 0.15, 1000, 16)      # batch 0 of bench.py
ctx = D.Ctx(idx, 0, max_read_len=0, max_batch_reads=0, input_slots=1)
ctx.use_synthetic_filter(mib << 20, 0.2)
ctx.upload_text(p, nb, off, ln, n)
for _ in range(3):
    ctx.run(); t = ctx.timing()
    print("seed lookup %.2f ms, %.3f probes/base" % (t.seed_probe_ms, t.windows / max(t.bases, 1)))
ctx.close()
