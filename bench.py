#!/usr/bin/env python3
"""bench.py -- `deSAMBA classify` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--reads-per-gpu R] [--read-len L] [--batches B]

Workload (BASELINE.json configs[1] shape): demo viral-gs index + synthetic 50 kbp ONT-15%-error reads, B DISTINCT
batches of R reads per GPU (default 16 x 65536 = 1 M reads, fewer if host memory is short; tools/readgen.c).

A step = one pass of the whole device path (encode -> exist-kmer probe -> classify) over one batch that is already
resident in HBM when the timed region starts; step i runs batch i mod B.  `value` = reads / s over the K timed steps.
`end_to_end` = the same B batches streamed from pinned host memory through two contexts of the GPU (they share the
staged index): H2D of the raw FASTQ text, all kernels, D2H of the results inside the timed region.
Reads are sharded over ranks (one process per GPU, index replicated, no data-path collective): weak scaling.
With --gpus N > 1 and no WORLD_SIZE in the environment this script starts the N ranks itself
(python -m torch.distributed.run), before anything touches a GPU.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step, `roofline_seed_lookup` the
seed-lookup kernel the north star names; both use ALGORITHMIC bytes -- for the classify kernel
from work counters counted on the device during the timed steps -- over the kernel's HIP-event time.
`cpu_baseline` times the unmodified reference binary (oracle/_ref/deSAMBA classify -t <cores>) on the first
--cpu-sample reads of batch 0 (median of 3 after a cache-warming run); `parity_sample` compares the SAM of those
reads: this GPU path vs the reference's UB-pinned build (must be 0 differing reads) and the stock reference vs the
UB-pinned build (the reference's own history-dependent residue).
The demo index itself is input data built by tools/make_demo_index.sh; nothing of oracle/ is on the measured path.
"""
import argparse
import ctypes as C
import json
import os
import re
import shutil
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.3 TB/s achievable)


def self_launch(a):
    """--gpus N without a torchrun environment: start the N ranks here, before any GPU call"""
    import torch
    rehearsal = bool(os.environ.get("DSB_BENCH_REHEARSAL"))
    n_dev = torch.cuda.device_count()          # does not initialise the GPU
    if not rehearsal and n_dev < a.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d GPUs, this machine has %d (DSB_BENCH_REHEARSAL=1 shares GPU 0 / rehearses the control flow)\n"
                         % (a.gpus, a.gpus, n_dev))
        sys.exit(2)
    port = 29000 + os.getpid() % 3000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


class Gen:
    """tools/readgen.c through ctypes"""

    def __init__(self, index_dir):
        so = os.path.join(ROOT, "tools", "libreadgen.so"); src = os.path.join(ROOT, "tools", "readgen.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lpthread", "-lm"])
        L = C.CDLL(so)
        L.readgen_open.argtypes = [C.c_char_p]; L.readgen_open.restype = C.c_long
        L.readgen_fill.argtypes = [C.c_long, C.c_void_p, C.c_size_t, C.c_long, C.c_long, C.c_double, C.c_uint64, C.c_int, C.c_int,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.readgen_fill.restype = C.c_long
        L.readgen_close.argtypes = [C.c_long]
        self.L = L
        self.h = L.readgen_open(os.fsencode(index_dir))
        if not self.h:
            raise RuntimeError("readgen_open(%s)" % index_dir)

    def fill(self, buf, cap, n, length, err, seed, threads):
        off = (C.c_uint64 * n)(); ln = (C.c_uint32 * n)()
        nb = self.L.readgen_fill(self.h, buf, cap, n, length, err, seed, 0, threads, off, ln)
        if nb < 0:
            raise RuntimeError("readgen_fill: buffer too small")
        return nb, off, ln


def run_ref(binary, index_dir, fq, out, threads):
    p = subprocess.run([binary, "classify", "-t", str(threads), index_dir, fq, "-o", out], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
    m = re.search(rb"(\d+) sequences processed in ([0-9.]+)s", p.stderr)
    if not m:
        sys.stderr.write("bench.py: %s exited with %d: %s\n" % (binary, p.returncode, p.stderr[-400:].decode(errors="replace")))
    return float(m.group(2)) if m else None


def sam_by_read(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, bytes) else open(path_or_bytes, "rb").read()
    d = {}
    for ln in data.splitlines():
        d.setdefault(ln.split(b"\t", 1)[0], []).append(ln)
    return d


def cpu_baseline(index_dir, sample_fq, n_sample, bases, gpu_sam):
    """reference binary on the host cores + parity of the sample"""
    ref = os.path.join(ROOT, "oracle", "_ref", "deSAMBA"); ubf = os.path.join(ROOT, "oracle", "_ref", "deSAMBA_ubfree")
    if not os.path.exists(ref):
        return None, None
    cores = len(os.sched_getaffinity(0))
    out = sample_fq + ".sam"
    run_ref(ref, index_dir, sample_fq, out, cores)               # warms the page cache (index, reads)
    secs = [run_ref(ref, index_dir, sample_fq, out, cores) for _ in range(3)]
    if any(s is None for s in secs):
        return None, None
    med = statistics.median(secs)
    base = {"value": n_sample / med, "unit": "reads/s", "gbp_per_s": bases / med / 1e9, "cores": cores, "kind": "reference",
            "seconds": secs,
            "sample": "first %d reads of batch 0 (same reads as the GPU run), `classify -t %d`, reference's own timer (index load excluded), median of 3 after one warming run" % (n_sample, cores)}
    parity = None
    if os.path.exists(ubf):
        stock = sam_by_read(out)
        ub_out = sample_fq + ".ub.sam"
        run_ref(ubf, index_dir, sample_fq, ub_out, cores)
        ub = sam_by_read(ub_out)
        gpu = sam_by_read(gpu_sam)
        parity = {"reads": n_sample,
                  "gpu_vs_ubpinned_differing_reads": sum(1 for k in ub if gpu.get(k) != ub[k]) + sum(1 for k in gpu if k not in ub),
                  "stock_vs_ubpinned_differing_reads": sum(1 for k in ub if stock.get(k) != ub[k]),
                  "stock_vs_ubpinned_flag_or_reference_changes": sum(1 for k in ub if [l.split(b"\t")[1:3] for l in stock.get(k, [])] != [l.split(b"\t")[1:3] for l in ub[k]]),
                  "note": "UB-pinned = the reference with its output-affecting undefined behaviour fixed to the canonical semantics (oracle/Makefile); the stock binary differs from it, and from itself at another thread count, in AS/POS/CIGAR digits only"}
        os.remove(ub_out)
    os.remove(out)
    return base, parity


def second_index(a, D, L, local, gen_threads):
    """A second, larger index made inside the bench: a synthetic strain collection (tools/synth_ref.py: base genomes,
    0-3 strains each at 0.5-4 % divergence, mobile elements, tandem repeats of unit length >= 12 -- no dinucleotide
    repeats) is indexed by THIS repo's builder on the GPU (dsb_index_build), then 50-kbp reads simulated from it are
    classified; a sample of the reads is compared with the reference binary (UB-pinned build) on the same index."""
    d = os.path.join(ROOT, "data", "bench_strain")
    shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    fa = os.path.join(d, "syn.fa"); idxd = os.path.join(d, "index")
    t0 = time.perf_counter()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, str(a.second_index_mbp), "11", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
    t_syn = time.perf_counter() - t0
    st = D.build_index(fa, idxd, device=local)
    os.remove(fa)
    t0 = time.perf_counter()
    idx2 = D.Index(idxd); gen2 = Gen(idxd)
    n = a.second_index_reads; Lr = 50000
    cap = n * (2 * Lr + 40) + (1 << 20)
    p = L.dsb_host_alloc(cap)
    nb, off, ln = gen2.fill(p, cap, n, Lr, 0.15, 777, gen_threads)
    ctx = D.Ctx(idx2, local, max_read_len=Lr, max_batch_reads=n)
    ctx.upload_text(p, nb, off, ln, n)
    t_open = time.perf_counter() - t0
    ms = []; tm = None
    for _ in range(4):
        ctx.run(); tm = ctx.timing(); ms.append(tm.total_ms)
    med = sorted(ms[1:])[1]
    res = ctx.fetch(strict=False)
    out = {"workload": "%d synthetic 50000 bp ONT-15%%-error reads on a %.0f-Mbp synthetic strain index (%d sequences, exist-k-mer length %d) built by this repo's GPU builder"
                       % (n, st.n_bases / 1e6, st.n_refs, L.dsb_index_ek_len(idx2.h)),
           "index_build": {"seconds": st.total_s, "stages_s": {"read_fasta": st.parse_s, "kmers_sort": st.sort_s, "graph": st.graph_s, "unitigs": st.walk_s, "bwt_rows": st.rows_s,
                                                               "tables_and_copy": st.tables_s, "write_files": st.write_s},
                           "bases": st.n_bases, "kmers_31": st.n_kmer, "unitigs": st.n_unitig, "bwt_rows": st.n_rows, "mbp_per_s": st.n_bases / 1e6 / st.total_s,
                           "reference_generation_s": t_syn},
           "open_stage_generate_s": t_open,
           "reads_per_s": n / (med / 1e3), "gbp_per_s": n * Lr / (med / 1e3) / 1e9, "ms_per_step": med,
           "kernel_ms": {"k_encode": tm.encode_ms, "order": tm.order_ms, "seed": tm.seed_probe_ms, "k_classify": tm.classify_ms, "tail": tm.tail_ms},
           "reads_in_second_run": tm.n_retry, "reads_mapped_frac": sum(1 for i in range(n) if res.reads[i].n > 0) / float(n),
           "reads_with_device_status": sum(1 for i in range(n) if res.reads[i].status != 0), "parity_sample": None}
    ubf = os.path.join(ROOT, "oracle", "_ref", "deSAMBA_ubfree")
    ns = min(a.second_index_parity_reads, n)
    if ns and os.path.exists(ubf):
        raw = C.string_at(p, off[ns - 1] + 2 * Lr + 8)
        names = []; pos = 0
        for i in range(ns):
            e = raw.index(b"\n", pos); names.append(raw[pos + 1:e]); pos = off[i] + 2 * ln[i] + 4
        reads = D.make_reads([(names[i], raw[off[i]:off[i] + ln[i]], None) for i in range(ns)])
        gpu = sam_by_read(D.format_sam(idx2, reads, res))
        sample = os.path.join(d, "sample.fq")
        with open(sample, "wb") as f:
            f.write(raw[:pos])
        secs = run_ref(ubf, idxd, sample, sample + ".sam", len(os.sched_getaffinity(0)))
        if secs is not None:
            ub = sam_by_read(sample + ".sam")
            out["parity_sample"] = {"reads": ns, "gpu_vs_ubpinned_differing_reads": sum(1 for k in ub if gpu.get(k) != ub[k]) + sum(1 for k in gpu if k not in ub),
                                    "reference_seconds": secs, "reference_reads_per_s": ns / secs}
    ctx.close(); idx2.close(); L.dsb_host_free(p)
    shutil.rmtree(d, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-gpu", type=int, default=65536, help="reads per batch (= per step) and GPU")
    ap.add_argument("--read-len", type=int, default=50000)
    ap.add_argument("--batches", type=int, default=0, help="distinct batches per GPU (0 = 16, fewer if host memory is short)")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-short-reads", action="store_true", help="skip the 1 M x 150 bp measurement (BASELINE configs[2] shape)")
    ap.add_argument("--no-seed-hbm", action="store_true", help="skip the seed-lookup measurement on synthetic multi-GiB filter tables")
    ap.add_argument("--no-second-index", action="store_true", help="skip building and measuring the second (synthetic strain) index")
    ap.add_argument("--second-index-mbp", type=int, default=320)
    ap.add_argument("--second-index-reads", type=int, default=65536)
    ap.add_argument("--second-index-parity-reads", type=int, default=256)
    ap.add_argument("--seed-hbm-mib", type=int, default=2048, help="size of each synthetic filter table (MiB, power of two 128 .. 16384)")
    ap.add_argument("--slots", type=int, default=0, help="reads in flight per GPU (0 = library default)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (a.gpus, world)); sys.exit(2)
    rehearsal = bool(os.environ.get("DSB_BENCH_REHEARSAL"))
    import torch
    have_gpu = torch.cuda.device_count() > 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL (backend "nccl").  DSB_BENCH_REHEARSAL=1: all ranks share GPU 0 (if there is one) and
        # rendezvous over gloo -- only to exercise the multi-process control flow on a box with fewer GPUs than ranks.
        if rehearsal:
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import __graft_entry__ as G
    import desamba_amd as D
    if rank == 0:
        if not os.path.exists(D.LIB_PATH):
            G.build()
        G.demo_dir()
    if dist:
        dist.barrier()
    index_dir = os.path.join(ROOT, "data", "demo", "index")

    if not have_gpu:
        if not rehearsal:
            raise SystemExit("bench.py: no GPU (there is no CPU path)")
        # control-flow rehearsal on a machine without a GPU: rendezvous, barrier, max-reduce, one JSON line; nothing is measured
        t = torch.tensor([0.0], dtype=torch.float64)
        if dist:
            dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "classified reads/s (50 kbp ONT reads, viral index)", "value": None, "unit": "reads/s", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "rehearsal": "control flow only: no GPU here, nothing was computed or measured"}), flush=True)
        if dist:
            dist.destroy_process_group()
        return

    R, Lr = a.reads_per_gpu, a.read_len
    rec_bytes = 2 * Lr + 40
    mem_avail = 64 << 30
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                mem_avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    ranks_here = world
    B = a.batches or max(1, min(16, int(mem_avail * 0.35 / ranks_here / (R * rec_bytes))))
    ncpu = len(os.sched_getaffinity(0))
    gen_threads = max(1, min(32, ncpu // ranks_here))

    idx = D.Index(index_dir)
    gen = Gen(index_dir)
    L = D.lib()
    t_gen0 = time.perf_counter()
    cap = R * rec_bytes + (1 << 20)
    bufs = []
    for b in range(B):
        p = L.dsb_host_alloc(cap)
        if not p:
            raise SystemExit("bench.py: cannot allocate %d bytes of pinned memory" % cap)
        nb, off, ln = gen.fill(p, cap, R, Lr, 0.15, 1000 * (rank + 1) + b, gen_threads)
        bufs.append((p, nb, off, ln))
    t_gen = time.perf_counter() - t_gen0
    bases_per_batch = R * Lr

    ctx = D.Ctx(idx, local, n_slots=a.slots, max_read_len=Lr, max_batch_reads=R, input_slots=B)
    for b in range(B):                          # all batches resident in HBM before any timed region
        ctx.select_slot(b); ctx.set_history(Lr if b else 0)
        ctx.upload_text(bufs[b][0], bufs[b][1], bufs[b][2], bufs[b][3], R)

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier(); torch.cuda.synchronize()

    if not rehearsal:
        torch.cuda.set_device(local)
    for i in range(a.warmup):
        ctx.select_slot(i % B); ctx.run()
    sync_all()
    t0 = time.perf_counter()
    seed_scan_used = False
    acc = {"probe": 0.0, "cls": 0.0, "enc": 0.0, "order": 0.0, "tail": 0.0, "windows": 0, "p1": 0, "bases": 0, "occ": 0, "mem": 0, "sa": 0, "rb": 0,
           "m_occ": 0, "m_mem": 0, "m_sa": 0, "m_rb": 0, "early": 0, "retry": 0}
    for i in range(a.steps):
        ctx.select_slot((a.warmup + i) % B)
        ctx.run()                               # launches all kernels of the step and synchronises the ctx's stream
        tm = ctx.timing()
        acc["probe"] += tm.seed_probe_ms; acc["cls"] += tm.classify_ms; acc["enc"] += tm.encode_ms; acc["order"] += tm.order_ms; acc["tail"] += tm.tail_ms
        acc["windows"] += tm.windows; acc["p1"] += tm.probes_t1; acc["bases"] += tm.bases
        acc["occ"] += tm.n_occ; acc["mem"] += tm.n_mem; acc["sa"] += tm.n_sa; acc["rb"] += tm.ref_bases
        acc["m_occ"] += tm.main_occ; acc["m_mem"] += tm.main_mem; acc["m_sa"] += tm.main_sa; acc["m_rb"] += tm.main_ref_bases
        acc["early"] += tm.n_early; acc["retry"] += tm.n_retry; seed_scan_used = bool(tm.seed_scan)
    sync_all()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # results of the last step's batch (statistics) and of batch 0 (parity of the CPU sample)
    res = ctx.fetch(strict=False)
    n_bad = sum(1 for i in range(R) if res.reads[i].status != 0)
    n_mapped = sum(1 for i in range(R) if res.reads[i].n > 0)
    dev_us = sorted(res.reads[i].device_us for i in range(R))

    # ---- end to end: the B batches streamed from pinned host memory through two contexts of this GPU -------------------
    e2e = None
    if not a.no_end_to_end:
        ctx2 = D.Ctx(idx, local, n_slots=a.slots, max_read_len=Lr, max_batch_reads=R, input_slots=1)
        ctx.select_slot(0)
        n_rounds = max(1, -(-8 // B))           # at least 8 batches in flight through the pipe
        order = [b for _ in range(n_rounds) for b in range(B)]
        lock = threading.Lock(); nxt = [0]; hits = [0, 0]; errs = []

        def worker(k, c):
            try:
                while True:
                    with lock:
                        j = nxt[0]; nxt[0] += 1
                    if j >= len(order):
                        return
                    b = order[j]
                    c.set_history(Lr if j else 0)
                    c.upload_text(bufs[b][0], bufs[b][1], bufs[b][2], bufs[b][3], R)
                    c.run()
                    r = c.fetch(strict=False)
                    hits[k] += r.n_hits
            except Exception as ex:          # noqa
                errs.append(ex)
        # one untimed batch per context first (nothing is allocated inside the timed region: the arenas come from the hints)
        for c in (ctx, ctx2):
            c.set_history(0); c.upload_text(bufs[0][0], bufs[0][1], bufs[0][2], bufs[0][3], R); c.run(); c.fetch(strict=False)
        sync_all()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(k, c)) for k, c in enumerate((ctx, ctx2))]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        sync_all()
        dt2 = time.perf_counter() - t0
        if errs:
            raise errs[0]
        if dist:
            t = torch.tensor([dt2], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        e2e = {"reads_per_s": len(order) * R * world / dt2, "gbp_per_s": len(order) * bases_per_batch * world / dt2 / 1e9, "ms_per_step": dt2 / len(order) * 1e3,
               "batches": len(order), "hits_fetched": hits[0] + hits[1],
               "what": "per batch: H2D of the raw FASTQ text (sequence + quality lines, %.1f GB) from pinned memory, all kernels, D2H of per-read results and hits; two contexts on the GPU share the staged index; the kernels of one batch at a time (dsb_batch_run takes a per-device turn), the other context's upload and fetch overlap them" % (bufs[0][1] / 1e9)}
        ctx2.close()

    # ---- the seed-lookup kernel in the HBM regime: 2 x 2 GiB synthetic filter tables, 20 % full (no index of that size exists here;
    # SURVEY.md 8d asks for the roofline claim on multi-GiB tables).  Only the seed lookup runs; answers on such tables are
    # checked against a host recomputation in tests/test_gpu_parity.py::test_seed_lookup_on_synthetic_multi_gib_tables.
    seed_hbm = None
    if rank == 0 and not a.no_seed_hbm:
        ctx3 = D.Ctx(idx, local, max_read_len=0, max_batch_reads=0, input_slots=1)
        ctx3.use_synthetic_filter(a.seed_hbm_mib << 20, 0.2)
        ctx3.upload_text(bufs[0][0], bufs[0][1], bufs[0][2], bufs[0][3], R)
        ms = []; tm3 = None
        for _ in range(4):
            ctx3.run(); tm3 = ctx3.timing(); ms.append(tm3.seed_probe_ms)
        ms = sorted(ms[1:])[len(ms[1:]) // 2]
        by = tm3.bases + 64.0 * (tm3.windows + tm3.probes_t1)
        seed_hbm = {"kernel": "k_seed_scan" if tm3.seed_scan else "k_seed_probe", "bound": "hbm", "tables": "2 x %d MiB synthetic, 20 %% of the bits set (k = 18)" % a.seed_hbm_mib,
                    "ms": ms, "algorithmic_bytes": by, "achieved": by / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                    "probes_per_base": tm3.windows / max(tm3.bases, 1), "table1_probes_per_base": tm3.probes_t1 / max(tm3.bases, 1), "traffic": None}
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_seed_hbm.json")))
            wl = prof.get("workload", {})
            if wl.get("reads_per_gpu") == R and wl.get("read_len") == Lr and wl.get("library") == D.lib().dsb_version().decode() and wl.get("table_mib") == a.seed_hbm_mib:
                cn = prof["counters"][seed_hbm["kernel"]]
                seed_hbm["traffic"] = (cn["FETCH_SIZE"] + cn["WRITE_SIZE"]) * 1024.0
        except Exception:
            pass
        ctx3.close()

    # ---- BASELINE configs[2] shape: 1 M synthetic 150 bp reads (1 % error) on the same index, one resident batch
    short = None
    if rank == 0 and not a.no_short_reads:
        n2 = 1 << 20; L2 = 150
        cap2 = n2 * (2 * L2 + 48) + (1 << 20)
        p2 = L.dsb_host_alloc(cap2)
        nb2, off2, ln2 = gen.fill(p2, cap2, n2, L2, 0.01, 4242, gen_threads)
        ctx4 = D.Ctx(idx, local, max_read_len=L2, max_batch_reads=n2)
        ctx4.upload_text(p2, nb2, off2, ln2, n2)
        ms = []; tm4 = None
        for _ in range(4):
            ctx4.run(); tm4 = ctx4.timing(); ms.append(tm4.total_ms)
        ms = sorted(ms[1:])[1]
        r4 = ctx4.fetch(strict=False)
        short = {"workload": "1048576 synthetic 150 bp reads, 1 % error, one batch resident in HBM", "reads_per_s": n2 / (ms / 1e3), "gbp_per_s": n2 * L2 / (ms / 1e3) / 1e9, "ms": ms,
                 "kernel_ms": {"k_encode": tm4.encode_ms, "order": tm4.order_ms, "seed": tm4.seed_probe_ms, "k_classify": tm4.classify_ms, "tail": tm4.tail_ms},
                 "reads_mapped_frac": sum(1 for i in range(0, n2, 64) if r4.reads[i].n > 0) / (n2 / 64.0)}
        ctx4.close(); L.dsb_host_free(p2)

    second = None
    if rank == 0 and not a.no_second_index:
        second = second_index(a, D, L, local, gen_threads)

    if rank == 0:
        steps = max(a.steps, 1)
        value = R * world * a.steps / dt
        gbp = acc["bases"] * world / dt / 1e9
        probe_s = acc["probe"] / steps / 1e3; classify_s = acc["cls"] / steps / 1e3
        # algorithmic bytes per launch, averaged over the timed steps
        seed_bytes = (acc["bases"] + 64.0 * (acc["windows"] + acc["p1"])) / steps
        # classify: 64 B per occ() (one rank line), 16 B per MEM search (a hash_index pair), 24 B per SA/unitig/ref-position lookup,
        # the 2-bit reference windows, and the 2 byte strands of the reads the launch handled; all counted on the device
        main_reads_frac = 1.0 - acc["early"] / float(R * steps)
        cls_bytes = (64.0 * acc["m_occ"] + 16.0 * acc["m_mem"] + 24.0 * acc["m_sa"] + acc["m_rb"] / 4.0 + 2.0 * acc["bases"] * main_reads_frac) / steps
        dom_is_cls = classify_s >= probe_s
        roof_seed = {"kernel": "k_seed_scan" if seed_scan_used else "k_seed_probe", "bound": "hbm", "achieved": seed_bytes / probe_s / 1e9 if probe_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "traffic": None, "ms": probe_s * 1e3, "algorithmic_bytes": seed_bytes}
        roof_seed["frac"] = roof_seed["achieved"] / HBM_PEAK_GBS
        roof_cls = {"kernel": "k_classify", "bound": "hbm", "achieved": cls_bytes / classify_s / 1e9 if classify_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "traffic": None, "ms": classify_s * 1e3, "algorithmic_bytes": cls_bytes,
                    "work_per_bp": {"occ": acc["occ"] / max(acc["bases"], 1), "mem_searches": acc["mem"] / max(acc["bases"], 1), "sa_lookups": acc["sa"] / max(acc["bases"], 1),
                                    "ref_bases": acc["rb"] / max(acc["bases"], 1)}}
        roof_cls["frac"] = roof_cls["achieved"] / HBM_PEAK_GBS
        # HBM-side traffic per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command (separate passes; KB ->
        # bytes; random 64-B gathers, so the guide's x2 correction for 128-B streaming requests does not apply), committed
        # under profiles/ -- quoted only when that profile was taken on this workload with this library build
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
            wl = prof.get("workload", {})
            if wl.get("reads_per_gpu") == R and wl.get("read_len") == Lr and wl.get("library") == D.lib().dsb_version().decode():
                cn = prof["counters"]
                roof_seed["traffic"] = (cn[roof_seed["kernel"]]["FETCH_SIZE"] + cn[roof_seed["kernel"]]["WRITE_SIZE"]) * 1024.0
                roof_cls["traffic"] = (cn["k_classify"]["FETCH_SIZE"] + cn["k_classify"]["WRITE_SIZE"]) * 1024.0
        except Exception:
            pass
        out = {
            "metric": "classified reads/s (50 kbp ONT reads, viral index)", "value": value, "unit": "reads/s", "gbp_per_s": gbp,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64 integer", "data": "synthetic",
            "config": {"workload": "demo viral-gs index (463 genomes, k=16 filter, 828 MB) + %d distinct batches of %d synthetic %d bp ONT-15%%-error reads per GPU (BASELINE configs[1] shape; %d reads per GPU)" % (B, R, Lr, B * R),
                       "reads_per_gpu": R, "read_len": Lr, "batches_per_gpu": B, "parallelism": "reads sharded x%d, index replicated" % world},
            "kernel_ms_per_step": {"k_encode": acc["enc"] / steps, "order+early_probe": acc["order"] / steps, ("k_seed_scan" if seed_scan_used else "k_seed_probe"): acc["probe"] / steps,
                                   "k_classify": acc["cls"] / steps, "wait_for_k_classify_early": acc["tail"] / steps},
            "reads_in_early_launch": acc["early"] / steps,
            "reads_in_second_run": acc["retry"] / steps,    # an arena or the loop budget outgrown; their time is part of the wait term
            "roofline": roof_cls if dom_is_cls else roof_seed,
            "roofline_seed_lookup": roof_seed,
            "roofline_seed_lookup_hbm": seed_hbm,
            "config2_short_reads": short,
            "second_index": second,
            "end_to_end": e2e,
            "reads_mapped_frac": n_mapped / max(R, 1), "reads_with_device_status": n_bad,
            "per_read_wave_us": {"mean": sum(dev_us) / max(len(dev_us), 1), "median": dev_us[len(dev_us) // 2] if dev_us else 0,
                                 "p99": dev_us[int(len(dev_us) * 0.99)] if dev_us else 0, "max": dev_us[-1] if dev_us else 0},
            "generation_s": t_gen,
        }
        if not a.no_cpu_baseline:
            ns = min(a.cpu_sample, R)
            ctx.select_slot(0); ctx.set_history(0)
            ctx.upload_text(bufs[0][0], bufs[0][1], bufs[0][2], bufs[0][3], R); ctx.run(); res0 = ctx.fetch(strict=False)
            # SAM of the sample, formatted by the library's writer from the results of batch 0
            raw = C.string_at(bufs[0][0], bufs[0][2][ns - 1] + 2 * Lr + 8)
            names = []
            pos = 0
            for i in range(ns):
                e = raw.index(b"\n", pos); names.append(raw[pos + 1:e]); pos = bufs[0][2][i] + 2 * bufs[0][3][i] + 4
            reads = D.make_reads([(names[i], raw[bufs[0][2][i]:bufs[0][2][i] + bufs[0][3][i]], None) for i in range(ns)])
            gpu_sam = D.format_sam(idx, reads, res0)
            tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
            sample = os.path.join(tmp, "dsb_bench_sample.fq")
            with open(sample, "wb") as f:
                f.write(raw[:pos])
            out["cpu_baseline"], out["parity_sample"] = cpu_baseline(index_dir, sample, ns, ns * Lr, gpu_sam)
            os.remove(sample)
            if e2e and out["cpu_baseline"]:
                e2e["vs_cpu_baseline"] = e2e["reads_per_s"] / world / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    for (p, _, _, _) in bufs:
        L.dsb_host_free(p)
    ctx.close(); idx.close()
    if dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
