#!/usr/bin/env python3
"""bench.py -- `deSAMBA classify` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--reads-per-gpu R] [--read-len L] [--batches B]

Headline workload (BASELINE.json configs[1]: "viral RefSeq index (~1 GB), synthetic 50 kbp ONT-15 %-error reads"): the RefSeq
index is not in the mount and cannot be downloaded, so a viral-RefSeq-sized one is made here -- a >= 300-Mbp synthetic strain
collection (tools/synth_ref.py: base genomes, 0-3 strains each at 0.5-4 % divergence, mobile elements, tandem repeats)
indexed by THIS repo's builder on the GPU (dsb_index_build, ~1 GB of index files) -- and B distinct batches of R reads per GPU
are simulated from it (tools/readgen.c).

A step = one pass of the whole device path (encode -> exist-kmer seed lookup -> classify) over one batch that is already
resident in HBM when the timed region starts; step i runs batch i mod B.  `value` = reads / s over the K timed steps.
`end_to_end` = batches streamed from pinned host memory through two contexts of the GPU (H2D, all kernels, D2H inside the
timed region).  `cli_end_to_end` = the product: the `deSAMBA classify` binary on a FASTQ file in /dev/shm, its own
"processed in" figure (reader, parser, upload, kernels, SAM writer), SAM compared with the library path; with --gpus N > 1
it is ONE process driving all N GPUs (`-g 0,..,N-1`) from one file -- the same reader code a user runs.
Reads are sharded over ranks (one process per GPU, index replicated, no data-path collective): weak scaling.
With --gpus N > 1 and no WORLD_SIZE in the environment this script starts the N ranks itself
(python -m torch.distributed.run), before anything touches a GPU.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step, `roofline_seed_lookup` the
seed-lookup kernel the north star names; both use ALGORITHMIC bytes -- for the classify kernel
from work counters counted on the device during the timed steps -- over the kernel's HIP-event time.
`cpu_baseline` times the unmodified reference binary (oracle/_ref/deSAMBA classify) on the first --cpu-sample reads of batch 0
(median of 3 after a cache-warming run, at the better of two thread counts: all visible CPUs / the cgroup's CPU quota);
`parity_sample` compares the SAM of those reads: this GPU path vs the reference's UB-pinned build (must be 0 differing reads),
the stock reference vs the UB-pinned build, and the stock reference with itself at -t 1 vs -t N (its own history-dependent
residue).  `demo_index` repeats the measurements on the reference's own demo index (463 viral
genomes, 11.5 Mbp: round 1-2's headline), where the device path is 4.5x faster and the host pipeline is what is tested.
Indexes and reads are input data; nothing of oracle/ is on a measured GPU path.
"""
import argparse
import ctypes as C
import hashlib
import datetime
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before the HIP runtime starts (the HIP runtime reads it at its first call)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.3 TB/s achievable)
CLI = os.path.join(ROOT, "desamba_amd", "bin", "deSAMBA")
REF = os.path.join(ROOT, "oracle", "_ref", "deSAMBA")
UBF = os.path.join(ROOT, "oracle", "_ref", "deSAMBA_ubfree")
SHM = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"


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


DEVICE_SOURCES = ["dsb_gpu.hip", "dsb_classify_dev.h", "dsb_wave.h", "dsb_device.h", "dsb_probe.h", "dsb_seed_scan.h"]


def device_source_md5():
    """md5 over the sources of the classify kernels: names the device code a PMC profile under profiles/ was taken on
    (tools/pmc_run.sh writes the same digest into the profile; a profile of other code is never quoted)"""
    h = hashlib.md5()
    for f in DEVICE_SOURCES:
        h.update(open(os.path.join(ROOT, "desamba_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def pmc_traffic(name, kernel, **match):
    """(bytes per launch, source) from profiles/<name> if it was taken on this device code and workload; else (None, why)"""
    path = os.path.join(ROOT, "profiles", name)
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profile %s" % name
    wl = prof.get("workload", {})
    if wl.get("device_source_md5") != device_source_md5():
        return None, "profiles/%s was taken on other device code (md5 %s, this build %s): not quoted" % (name, wl.get("device_source_md5"), device_source_md5())
    for k, v in match.items():
        if wl.get(k) != v:
            return None, "profiles/%s: %s = %r, this run %r: not quoted" % (name, k, wl.get(k), v)
    cn = prof["counters"].get(kernel)
    if not cn or "FETCH_SIZE" not in cn or "WRITE_SIZE" not in cn:
        return None, "profiles/%s holds no FETCH_SIZE / WRITE_SIZE of %s" % (name, kernel)
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own passes; KB -> bytes; random 64-byte gathers, so the guide's x2 correction
    # for 128-byte streaming requests does not apply
    return (cn["FETCH_SIZE"] + cn["WRITE_SIZE"]) * 1024.0, "profiles/%s (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes, per launch; device source md5 %s matches this build)" % (name, device_source_md5())


def gather_ceiling(table_mib):
    """tools/gather_bench on a table of that size, here and now on this GPU: what random 64-byte lines reach (GB/s), or None"""
    exe = os.path.join(ROOT, "tools", "gather_bench")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, str(table_mib)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout.decode()
        v = [float(m) for m in re.findall(r"=\s*([0-9.]+) GB/s", out)]
        return max(v) if v else None
    except (OSError, subprocess.SubprocessError):
        return None


def host_cpus():
    """CPUs this process may run on, and the CPU quota of its control group (None: unlimited)"""
    n = len(os.sched_getaffinity(0)); quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / float(p)
    except (OSError, ValueError):
        pass
    return n, quota


def mem_limit():
    lim = 64 << 30
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                lim = int(ln.split()[1]) * 1024
    except OSError:
        pass
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            lim = min(lim, int(v) - cur)
    except (OSError, ValueError):
        pass
    return lim


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

    def fill(self, buf, cap, n, length, err, seed, threads, profile=0):
        """profile 0: reads of one length (ONT / NGS error mix); 1: PacBio-mixed (log-normal lengths 500 .. 80000, `length` ignored)"""
        off = (C.c_uint64 * n)(); ln = (C.c_uint32 * n)()
        nb = self.L.readgen_fill(self.h, buf, cap, n, length, err, seed, profile, threads, off, ln)
        if nb < 0:
            raise RuntimeError("readgen_fill: buffer too small")
        return nb, off, ln

    def close(self):
        self.L.readgen_close(self.h); self.h = 0


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


def n_differ(a, b):
    return sum(1 for k in b if a.get(k) != b[k]) + sum(1 for k in a if k not in b)


def cpu_baseline(index_dir, sample_fq, n_sample, bases, gpu_sam, t1_reads):
    """reference binary on the host cores + parity of the sample"""
    if not os.path.exists(REF):
        return None, None
    cores, quota = host_cpus()
    out = sample_fq + ".sam"
    run_ref(REF, index_dir, sample_fq, out, cores)               # warms the page cache (index, reads)
    # the container may show more CPUs than its control group lets it use: the reference is timed at the better of the two thread counts
    cand = [cores] + ([max(1, int(round(quota)))] if quota and int(round(quota)) < cores else [])
    first = {t: run_ref(REF, index_dir, sample_fq, out, t) for t in cand}
    if any(v is None for v in first.values()):
        return None, None
    best = min(first, key=first.get)
    secs = [first[best]] + [run_ref(REF, index_dir, sample_fq, out, best) for _ in range(2)]
    if any(s is None for s in secs):
        return None, None
    med = statistics.median(secs)
    base = {"value": n_sample / med, "unit": "reads/s", "gbp_per_s": bases / med / 1e9, "cores": best, "kind": "reference",
            "host": {"cpus_visible": cores, "cgroup_cpu_quota": quota}, "seconds": secs, "seconds_first_run_by_threads": {str(k): v for k, v in first.items()},
            "sample": "first %d reads of batch 0 (same reads as the GPU run), `classify -t %d`, reference's own timer (index load excluded), median of 3 after one warming run" % (n_sample, best)}
    parity = None
    if os.path.exists(UBF):
        stock = sam_by_read(out)
        ub_out = sample_fq + ".ub.sam"
        run_ref(UBF, index_dir, sample_fq, ub_out, best)
        ub = sam_by_read(ub_out)
        gpu = sam_by_read(gpu_sam)
        parity = {"reads": n_sample,
                  "gpu_vs_ubpinned_differing_reads": n_differ(gpu, ub),
                  "stock_vs_ubpinned_differing_reads": sum(1 for k in ub if stock.get(k) != ub[k]),
                  "stock_vs_ubpinned_flag_or_reference_changes": sum(1 for k in ub if [l.split(b"\t")[1:3] for l in stock.get(k, [])] != [l.split(b"\t")[1:3] for l in ub[k]]),
                  "note": "UB-pinned = the reference with its output-affecting undefined behaviour fixed to the canonical semantics (oracle/Makefile); the stock binary differs from it, and from itself at another thread count, in AS/POS/CIGAR digits only"}
        os.remove(ub_out)
        if t1_reads:
            # the reference against itself: the first t1_reads reads at -t 1 and at -t N (a single thread is ~1/16 of the box: a small sample)
            t1 = min(t1_reads, n_sample)
            sub = sample_fq + ".t1.fq"
            with open(sample_fq, "rb") as f, open(sub, "wb") as g:
                for _ in range(4 * t1):
                    g.write(f.readline())
            s1 = run_ref(REF, index_dir, sub, sub + ".1.sam", 1); sn = run_ref(REF, index_dir, sub, sub + ".n.sam", best)
            if s1 is not None and sn is not None:
                a1 = sam_by_read(sub + ".1.sam"); an = sam_by_read(sub + ".n.sam")
                parity["stock_t1_vs_tN"] = {"reads": t1, "threads_N": best, "differing_reads": n_differ(a1, an),
                                            "flag_or_reference_changes": sum(1 for k in an if [l.split(b"\t")[1:3] for l in a1.get(k, [])] != [l.split(b"\t")[1:3] for l in an[k]]),
                                            "gpu_vs_ubpinned_on_these_reads": sum(1 for k in a1 if gpu.get(k) != ub.get(k)),
                                            "seconds_t1": s1}
            for x in (sub, sub + ".1.sam", sub + ".n.sam"):
                if os.path.exists(x):
                    os.remove(x)
    os.remove(out)
    return base, parity


def oracle_probe_counts(index_dir, fq_text, n_reads=48):
    """What the reference's scan consumes (SURVEY.md 8d's definition of the seed lookup's algorithmic bytes): P0 = windows probed in
    filter table 0, P1 = probes that go on to table 1, counted by the CPU restatement (oracle/) on the first reads of the CPU
    sample -- the checker beside the measurement, never on a GPU path.  -> {p0_per_base, p1_per_base, reads} or None"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        ora = oracle_lib.Oracle(index_dir)
    except Exception:
        return None
    lines = fq_text.split(b"\n"); p0 = p1 = bases = n = 0; hist = 0
    for i in range(1, min(len(lines), 4 * n_reads), 4):
        seq = lines[i]
        ora.classify(seq, hist); hist = max(hist, len(seq))
        c = ora.counters(); p0 += c[0]; p1 += c[1]; bases += len(seq); n += 1
    return {"p0_per_base": p0 / max(bases, 1), "p1_per_base": p1 / max(bases, 1), "reads": n} if n else None


def cli_end_to_end(index_dir, gen_args, n_reads, devices, lib_sam_md5=None, lib_n=0, reps=2):
    """the product: `deSAMBA classify` on a FASTQ file in /dev/shm; its own "processed in" figure"""
    fq = os.path.join(SHM, "dsb_bench_cli.fq"); sam = os.path.join(SHM, "dsb_bench_cli.sam")
    t0 = time.perf_counter()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fastq.py"), index_dir, fq, str(n_reads)] + [str(x) for x in gen_args], check=True, stdout=subprocess.DEVNULL)
    t_gen = time.perf_counter() - t0
    size = os.path.getsize(fq)
    runs = []; trace = []
    env = dict(os.environ); env["DSB_CLI_TRACE"] = "1"
    for _ in range(reps):
        t0 = time.perf_counter()
        p = subprocess.run([CLI, "classify", "-g", ",".join(str(d) for d in devices), index_dir, fq, "-o", sam], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, env=env)
        wall = time.perf_counter() - t0
        m = re.search(rb"(\d+) sequences processed in ([0-9.]+)s", p.stderr)
        if p.returncode != 0 or not m or int(m.group(1)) != n_reads:
            sys.stderr.write("bench.py: the CLI exited with %d: %s\n" % (p.returncode, p.stderr[-600:].decode(errors="replace")))
            os.remove(fq)
            return None
        runs.append((float(m.group(2)), wall))
        trace = [l.decode(errors="replace") for l in p.stderr.splitlines() if l.startswith(b"[trace]")]
    best = min(runs)
    out = {"reads": n_reads, "reads_per_s": n_reads / best[0], "gbp_per_s": size / 2.0 / best[0] / 1e9, "seconds": [r[0] for r in runs], "wall_seconds_incl_index_load": [r[1] for r in runs],
           "input": "%.1f GB of plain FASTQ in %s (generated in %.1f s)" % (size / 1e9, SHM, t_gen), "devices": list(devices),
           "what": "the `deSAMBA classify` binary, default options, SAM to a file in %s: mapped input parsed on all host threads, sequence lines gathered through pinned chunks, two contexts per GPU, ordered SAM writer; its own timer (index load excluded, as the reference's)" % SHM,
           "trace": trace}
    if lib_sam_md5:
        # the first lib_n reads of the file are batch 0 of the library path: same SAM bytes
        h = hashlib.md5(); k = 0
        with open(sam, "rb") as f:
            prev = None
            for ln in f:
                nm = ln.split(b"\t", 1)[0]
                if nm != prev:
                    k += 1; prev = nm
                    if k > lib_n:
                        break
                h.update(ln)
        out["sam_md5_equals_library_path"] = (h.hexdigest() == lib_sam_md5); out["sam_compared_reads"] = lib_n
    os.remove(fq); os.remove(sam)
    return out


class Measure:
    """B batches of R reads simulated from one index, resident in HBM; the timed steps; the streamed (end to end) run"""

    def __init__(self, D, L, idx, index_dir, local, a, R, Lr, B, seed0, gen_threads):
        self.D, self.L, self.idx, self.index_dir, self.local, self.a, self.R, self.Lr, self.B = D, L, idx, index_dir, local, a, R, Lr, B
        gen = Gen(index_dir)
        t0 = time.perf_counter()
        cap = R * (2 * Lr + 40) + (1 << 20)
        self.bufs = []
        for b in range(B):
            p = L.dsb_host_alloc(cap)
            if not p:
                raise SystemExit("bench.py: cannot allocate %d bytes of pinned memory" % cap)
            nb, off, ln = gen.fill(p, cap, R, Lr, 0.15, seed0 + b, gen_threads)
            self.bufs.append((p, nb, off, ln))
        gen.close()
        self.t_gen = time.perf_counter() - t0
        self.ctx = D.Ctx(idx, local, n_slots=a.slots, max_read_len=Lr, max_batch_reads=R, input_slots=B)
        for b in range(B):                          # all batches resident in HBM before any timed region
            self.ctx.select_slot(b); self.ctx.set_history(Lr if b else 0)
            self.ctx.upload_text(*self.bufs[b], R)

    def steps(self, warmup, steps, sync_all, reduce_max):
        ctx, B = self.ctx, self.B
        for i in range(warmup):
            ctx.select_slot(i % B); ctx.run()
        sync_all()
        t0 = time.perf_counter()
        acc = {"probe": 0.0, "cls": 0.0, "enc": 0.0, "order": 0.0, "tail": 0.0, "windows": 0, "p1": 0, "bases": 0, "occ": 0, "mem": 0, "sa": 0, "rb": 0,
               "m_occ": 0, "m_mem": 0, "m_sa": 0, "m_rb": 0, "early": 0, "retry": 0, "requeue": 0, "scan": False}
        for i in range(steps):
            ctx.select_slot((warmup + i) % B)
            ctx.run()                               # launches all kernels of the step and synchronises the ctx's stream
            tm = ctx.timing()
            acc["probe"] += tm.seed_probe_ms; acc["cls"] += tm.classify_ms; acc["enc"] += tm.encode_ms; acc["order"] += tm.order_ms; acc["tail"] += tm.tail_ms
            acc["windows"] += tm.windows; acc["p1"] += tm.probes_t1; acc["bases"] += tm.bases
            acc["occ"] += tm.n_occ; acc["mem"] += tm.n_mem; acc["sa"] += tm.n_sa; acc["rb"] += tm.ref_bases
            acc["m_occ"] += tm.main_occ; acc["m_mem"] += tm.main_mem; acc["m_sa"] += tm.main_sa; acc["m_rb"] += tm.main_ref_bases
            acc["early"] += tm.n_early; acc["retry"] += tm.n_retry; acc["requeue"] += tm.n_requeue; acc["scan"] = bool(tm.seed_scan)
        sync_all()
        dt = reduce_max(time.perf_counter() - t0)
        self.acc, self.dt, self.n_steps = acc, dt, steps
        res = ctx.fetch(strict=False)
        R = self.R
        self.n_bad = sum(1 for i in range(R) if res.reads[i].status != 0)
        self.n_mapped = sum(1 for i in range(R) if res.reads[i].n > 0)
        self.dev_us = sorted(res.reads[i].device_us for i in range(R))
        return dt

    def rooflines(self, profile_tag):
        """algorithmic bytes per launch, averaged over the timed steps"""
        acc, steps, R = self.acc, max(self.n_steps, 1), self.R
        probe_s = acc["probe"] / steps / 1e3; classify_s = acc["cls"] / steps / 1e3
        seed_bytes = (acc["bases"] + 64.0 * (acc["windows"] + acc["p1"])) / steps
        # classify: 64 B per occ() (one rank line), 16 B per MEM search (a hash_index pair), 24 B per SA/unitig/ref-position lookup,
        # the 2-bit reference windows, and the 2 byte strands of the reads the launch handled; all counted on the device
        main_reads_frac = 1.0 - acc["early"] / float(R * steps)
        cls_bytes = (64.0 * acc["m_occ"] + 16.0 * acc["m_mem"] + 24.0 * acc["m_sa"] + acc["m_rb"] / 4.0 + 2.0 * acc["bases"] * main_reads_frac) / steps
        kseed = "k_seed_scan" if acc["scan"] else "k_seed_probe"
        roof_seed = {"kernel": kseed, "bound": "hbm", "achieved": seed_bytes / probe_s / 1e9 if probe_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "traffic": None, "ms": probe_s * 1e3, "algorithmic_bytes": seed_bytes}
        roof_seed["frac"] = roof_seed["achieved"] / HBM_PEAK_GBS
        roof_cls = {"kernel": "k_classify", "bound": "hbm", "achieved": cls_bytes / classify_s / 1e9 if classify_s > 0 else 0.0, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "traffic": None, "ms": classify_s * 1e3, "algorithmic_bytes": cls_bytes,
                    "work_per_bp": {"occ": acc["occ"] / max(acc["bases"], 1), "mem_searches": acc["mem"] / max(acc["bases"], 1), "sa_lookups": acc["sa"] / max(acc["bases"], 1),
                                    "ref_bases": acc["rb"] / max(acc["bases"], 1)}}
        roof_cls["frac"] = roof_cls["achieved"] / HBM_PEAK_GBS
        # HBM-side traffic per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command, committed under profiles/ --
        # quoted only when that profile was taken on this workload with THIS device code (md5 of the kernel sources)
        roof_seed["traffic"], roof_seed["traffic_source"] = pmc_traffic("r04_pmc_traffic_%s.json" % profile_tag, kseed, reads_per_gpu=R, read_len=self.Lr)
        roof_cls["traffic"], roof_cls["traffic_source"] = pmc_traffic("r04_pmc_traffic_%s.json" % profile_tag, "k_classify", reads_per_gpu=R, read_len=self.Lr)
        return roof_cls, roof_seed, classify_s >= probe_s

    def kernel_ms(self):
        acc, steps = self.acc, max(self.n_steps, 1)
        return {"k_encode": acc["enc"] / steps, "order+early_probe": acc["order"] / steps, ("k_seed_scan" if acc["scan"] else "k_seed_probe"): acc["probe"] / steps,
                "k_classify": acc["cls"] / steps, "wait_for_early_and_heavy_launches": acc["tail"] / steps}

    def end_to_end(self, sync_all, reduce_max, world, min_batches=8):
        """the B batches streamed from pinned host memory through two contexts of this GPU"""
        D, a, R, Lr, B, bufs, ctx = self.D, self.a, self.R, self.Lr, self.B, self.bufs, self.ctx
        ctx2 = D.Ctx(self.idx, self.local, n_slots=a.slots, max_read_len=Lr, max_batch_reads=R, input_slots=1)
        ctx.select_slot(0)
        n_rounds = max(1, -(-min_batches // B))
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
                    c.upload_text(*bufs[b], R)
                    c.run()
                    r = c.fetch(strict=False)
                    hits[k] += r.n_hits
            except Exception as ex:          # noqa
                errs.append(ex)
        # one untimed batch per context first (nothing is allocated inside the timed region: the arenas come from the hints)
        for c in (ctx, ctx2):
            c.set_history(0); c.upload_text(*bufs[0], R); c.run(); c.fetch(strict=False)
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
        dt2 = reduce_max(dt2)
        ctx2.close()
        return {"reads_per_s": len(order) * R * world / dt2, "gbp_per_s": len(order) * R * Lr * world / dt2 / 1e9, "ms_per_step": dt2 / len(order) * 1e3,
                "batches": len(order), "hits_fetched": hits[0] + hits[1],
                "what": "per batch: H2D of the raw FASTQ text (sequence + quality lines, %.1f GB) from pinned memory, all kernels, D2H of per-read results and hits; two contexts on the GPU share the staged index; the kernels of one batch at a time (dsb_batch_run takes a per-device turn), the other context's upload and fetch overlap them" % (bufs[0][1] / 1e9)}

    def sample(self, ns):
        """SAM of the first ns reads of batch 0 (library's writer) and the FASTQ text of those reads"""
        D, ctx, bufs, Lr, R = self.D, self.ctx, self.bufs, self.Lr, self.R
        ctx.select_slot(0); ctx.set_history(0)
        ctx.upload_text(*bufs[0], R); ctx.run(); res0 = ctx.fetch(strict=False)
        raw = C.string_at(bufs[0][0], bufs[0][2][ns - 1] + 2 * Lr + 8)
        names = []; pos = 0
        for i in range(ns):
            e = raw.index(b"\n", pos); names.append(raw[pos + 1:e]); pos = bufs[0][2][i] + 2 * bufs[0][3][i] + 4
        reads = D.make_reads([(names[i], raw[bufs[0][2][i]:bufs[0][2][i] + bufs[0][3][i]], None) for i in range(ns)])
        return D.format_sam(self.idx, reads, res0), raw[:pos]

    def close(self):
        for (p, _, _, _) in self.bufs:
            self.L.dsb_host_free(p)
        self.bufs = []
        self.ctx.close()


def resident_batch(D, L, idx, index_dir, local, n, length, err, seed, profile, threads, workload, parity_reads=0):
    """One batch of n synthetic reads simulated from the index, resident in HBM: the whole device path four times, the median of the
    last three.  parity_reads > 0: the first so many reads also through the reference's UB-pinned build on the same index directory
    (SAM compared read by read) and through the stock build (timed: the CPU figure beside it)."""
    max_l = 80000 if profile == 1 else length
    cap = n * (2 * max_l + 48) + (1 << 20) if profile == 0 else n * 2 * 13500 + n * 48 + (64 << 20)
    p = L.dsb_host_alloc(cap)
    gen = Gen(index_dir)
    try:
        nb, off, ln = gen.fill(p, cap, n, length, err, seed, threads, profile)
    finally:
        gen.close()
    longest = max(ln[i] for i in range(0, n, max(1, n // 4096))) if profile == 1 else length
    ctx = D.Ctx(idx, local, max_read_len=(80000 if profile == 1 else length), max_batch_reads=n)
    ctx.upload_text(p, nb, off, ln, n)
    ms = []; tm = None
    for _ in range(4):
        ctx.run(); tm = ctx.timing(); ms.append(tm.total_ms)
    med = sorted(ms[1:])[1]
    res = ctx.fetch(strict=False)
    bases = tm.bases
    out = {"workload": workload, "reads": n, "bases": bases, "reads_per_s": n / (med / 1e3), "gbp_per_s": bases / (med / 1e3) / 1e9, "ms": med,
           "kernel_ms": {"k_encode": tm.encode_ms, "order": tm.order_ms, "seed": tm.seed_probe_ms, "k_classify": tm.classify_ms, "tail": tm.tail_ms},
           "reads_mapped_frac": sum(1 for i in range(0, n, 64) if res.reads[i].n > 0) / (len(range(0, n, 64)) or 1),
           "reads_with_device_status": sum(1 for i in range(n) if res.reads[i].status), "longest_read_sampled": longest}
    if parity_reads and os.path.exists(UBF):
        ns = min(parity_reads, n)
        raw = C.string_at(p, off[ns - 1] + 2 * ln[ns - 1] + 8)
        names = []; pos = 0
        for i in range(ns):
            e = raw.index(b"\n", pos); names.append(raw[pos + 1:e]); pos = off[i] + 2 * ln[i] + 4
        reads = D.make_reads([(names[i], raw[off[i]:off[i] + ln[i]], None) for i in range(ns)])
        gpu = sam_by_read(D.format_sam(idx, reads, res))
        fq = os.path.join(SHM, "dsb_bench_c5.fq")
        with open(fq, "wb") as f:
            f.write(raw[:pos])
        cores, quota = host_cpus(); t = max(1, int(round(quota))) if quota and int(round(quota)) < cores else cores
        run_ref(UBF, index_dir, fq, fq + ".ub.sam", t)                    # (also warms the page cache for the timed stock run)
        ub = sam_by_read(fq + ".ub.sam")
        sec = run_ref(REF, index_dir, fq, fq + ".ref.sam", t) if os.path.exists(REF) else None
        stock = sam_by_read(fq + ".ref.sam") if sec is not None else {}
        out["parity_sample"] = {"reads": ns, "gpu_vs_ubpinned_differing_reads": n_differ({k: gpu.get(k) for k in ub}, ub),
                                "stock_vs_ubpinned_differing_reads": sum(1 for k in ub if stock.get(k) != ub[k]) if stock else None,
                                "mapped_reads": sum(1 for k in ub if ub[k][0].split(b"\t")[1] != b"4")}
        if sec:
            out["cpu_baseline"] = {"value": ns / sec, "unit": "reads/s", "cores": t, "kind": "reference", "seconds": sec,
                                   "sample": "first %d reads of the batch, `classify -t %d`, the reference's own timer, one run after a warming run of the UB-pinned build" % (ns, t)}
        for x in (fq, fq + ".ub.sam", fq + ".ref.sam"):
            if os.path.exists(x):
                os.remove(x)
    ctx.close(); L.dsb_host_free(p)
    return out


def proxy_index(a, D, local):
    """BASELINE configs[4] stands for a bacterial-scale index that is not in the mount: its proxy is a synthetic collection of
    a.proxy_mbp million bases indexed here by this repo's builder (k = 17 / 18 filter tables, natural 64-bit rank counts from 4.3 G rows on)"""
    d = os.path.join(ROOT, "data", "bench_proxy"); idxd = os.path.join(d, "index")
    shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    fa = os.path.join(d, "syn.fa")
    t0 = time.perf_counter()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, str(a.proxy_mbp), "13", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
    t_syn = time.perf_counter() - t0
    st = D.build_index(fa, idxd, device=local)
    os.remove(fa)
    files = sum(os.path.getsize(os.path.join(idxd, f)) for f in os.listdir(idxd))
    return idxd, {"seconds": st.total_s, "bases": st.n_bases, "sequences": st.n_refs, "kmers_31": st.n_kmer, "unitigs": st.n_unitig, "bwt_rows": st.n_rows,
                  "index_files_bytes": files, "filter_table_bytes": os.path.getsize(os.path.join(idxd, "deSAMBA.exk0")), "reference_generation_s": t_syn}


def strain_index(a, D, rank, local, barrier):
    """the viral-RefSeq-sized index of the headline: made by rank 0 with this repo's builder, opened by every rank"""
    d = os.path.join(ROOT, "data", "bench_strain"); idxd = os.path.join(d, "index"); info = None
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        fa = os.path.join(d, "syn.fa")
        t0 = time.perf_counter()
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "synth_ref.py"), fa, str(a.index_mbp), "11", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
        t_syn = time.perf_counter() - t0
        st = D.build_index(fa, idxd, device=local)
        files = sum(os.path.getsize(os.path.join(idxd, f)) for f in os.listdir(idxd))
        # the same index once more within an eighth of the device memory the one-piece build held (passes over ranges of k-mer prefixes,
        # the k-mer list in a temporary file): the ten files must come out byte for byte
        budget_info = None
        if not a.no_budget_build:
            d2 = os.path.join(d, "index_budget"); budget = st.peak_device_bytes // 8
            os.environ["DSB_BUILD_BUDGET"] = str(budget); os.environ["DSB_BUILD_SPILL"] = "1"
            try:
                s2 = D.build_index(fa, d2, device=local)
            finally:
                del os.environ["DSB_BUILD_BUDGET"]; del os.environ["DSB_BUILD_SPILL"]

            def md5(p_):
                h = hashlib.md5()
                with open(p_, "rb") as f_:
                    for blk in iter(lambda: f_.read(1 << 24), b""):
                        h.update(blk)
                return h.hexdigest()
            same = sorted(os.listdir(idxd)) == sorted(os.listdir(d2)) and all(md5(os.path.join(idxd, f)) == md5(os.path.join(d2, f)) for f in os.listdir(idxd))
            budget_info = {"seconds": s2.total_s, "budget_bytes": s2.budget_bytes, "peak_device_bytes": s2.peak_device_bytes, "one_piece_peak_device_bytes": st.peak_device_bytes,
                           "passes": {"kmers": s2.ranges_kmers, "unitig_numbers": s2.ranges_unitig_numbers, "bwt_rows": s2.ranges_rows, "filter_tables": s2.ranges_exist},
                           "kmer_list_spilled_bytes": s2.spilled_bytes, "files_identical_to_the_one_piece_build": bool(same)}
            shutil.rmtree(d2, ignore_errors=True)
        os.remove(fa)
        info = {"seconds": st.total_s, "stages_s": {"read_fasta": st.parse_s, "kmers_sort": st.sort_s, "graph": st.graph_s, "unitigs": st.walk_s, "bwt_rows": st.rows_s,
                                                    "tables_and_copy": st.tables_s, "write_files": st.write_s},
                "bases": st.n_bases, "sequences": st.n_refs, "kmers_31": st.n_kmer, "unitigs": st.n_unitig, "bwt_rows": st.n_rows, "mbp_per_s": st.n_bases / 1e6 / st.total_s,
                "index_files_bytes": files, "reference_generation_s": t_syn, "peak_device_bytes": st.peak_device_bytes, "within_an_eighth_of_the_memory": budget_info}
    barrier()
    return idxd, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-gpu", type=int, default=65536, help="reads per batch (= per step) and GPU")
    ap.add_argument("--read-len", type=int, default=50000)
    ap.add_argument("--batches", type=int, default=0, help="distinct batches per GPU (0 = 4, fewer if host memory is short)")
    ap.add_argument("--index-mbp", type=int, default=320, help="size of the synthetic strain collection of the headline index")
    ap.add_argument("--headline", choices=["strain", "demo"], default="strain", help="profiling runs: `demo` puts the reference's demo index in the headline position (nothing is built)")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--t1-sample", type=int, default=1024, help="reads of the stock reference's -t 1 vs -t N self-comparison (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-cli", action="store_true", help="skip the runs of the deSAMBA binary (cli_end_to_end)")
    ap.add_argument("--cli-reads", type=int, default=262144, help="reads per GPU of the CLI run on the headline index")
    ap.add_argument("--no-demo-index", action="store_true", help="skip the measurements on the reference's demo index")
    ap.add_argument("--demo-batches", type=int, default=8)
    ap.add_argument("--demo-steps", type=int, default=8)
    ap.add_argument("--demo-cli-reads", type=int, default=524288)
    ap.add_argument("--no-short-reads", action="store_true", help="skip the 1 M x 150 bp measurement (BASELINE configs[2] shape)")
    ap.add_argument("--no-seed-hbm", action="store_true", help="skip the seed-lookup measurement on synthetic multi-GiB filter tables")
    ap.add_argument("--seed-hbm-mib", type=int, default=2048, help="size of each synthetic filter table (MiB, power of two 128 .. 16384)")
    ap.add_argument("--no-budget-build", action="store_true", help="skip building the headline index a second time within an eighth of the device memory (files compared)")
    ap.add_argument("--no-proxy", action="store_true", help="skip the BASELINE configs[4] proxy (a >= 1-Gbp index built in the run, PacBio-mixed reads)")
    ap.add_argument("--proxy-mbp", type=int, default=1000)
    ap.add_argument("--proxy-reads", type=int, default=65536)
    ap.add_argument("--proxy-parity", type=int, default=2048, help="reads of the proxy's parity sample against the reference's UB-pinned build")
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
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=30))   # rank 0 works alone for minutes at the end (CPU baseline, CLI run)

    import __graft_entry__ as G
    import desamba_amd as D
    if rank == 0:
        if not os.path.exists(D.LIB_PATH):
            G.build()
        G.demo_dir()

    def barrier():
        if dist:
            dist.barrier()
    barrier()
    demo_dir = os.path.join(ROOT, "data", "demo", "index")

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

    if not rehearsal:
        torch.cuda.set_device(local)

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier(); torch.cuda.synchronize()

    def reduce_max(v):
        if not dist:
            return v
        t = torch.tensor([v], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    L = D.lib()
    R, Lr = a.reads_per_gpu, a.read_len
    rec_bytes = 2 * Lr + 40
    ncpu, quota = host_cpus()
    gen_threads = max(1, min(32, int((quota or ncpu)) // world))
    B = a.batches or max(1, min(4, int(mem_limit() * 0.3 / world / (R * rec_bytes))))

    # ---- the headline: a viral-RefSeq-sized index built here, B batches per GPU simulated from it ----------------------
    if a.headline == "demo":
        index_dir, build_info = demo_dir, {"bases": 11476226, "sequences": 463, "kmers_31": 10982489, "index_files_bytes": 828e6, "seconds": 0.0}
    else:
        index_dir, build_info = strain_index(a, D, rank, local, barrier)
    idx = D.Index(index_dir)
    t0 = time.perf_counter()
    m = Measure(D, L, idx, index_dir, local, a, R, Lr, B, 1000 * (rank + 1), gen_threads)
    t_stage = time.perf_counter() - t0
    dt = m.steps(a.warmup, a.steps, sync_all, reduce_max)
    e2e = None if a.no_end_to_end else m.end_to_end(sync_all, reduce_max, world, min_batches=max(4, B))
    out = None
    if rank == 0:
        steps = max(a.steps, 1)
        roof_cls, roof_seed, dom_is_cls = m.rooflines(a.headline)
        idx_desc = "%.0f-Mbp synthetic strain collection (%d sequences, %d M 31-mers, %.2f GB of index files, exist-k-mer length %d) indexed here by dsb_index_build in %.1f s" % (
            build_info["bases"] / 1e6, build_info["sequences"], build_info["kmers_31"] / 1e6, build_info["index_files_bytes"] / 1e9, L.dsb_index_ek_len(idx.h), build_info["seconds"])
        out = {
            "metric": "classified reads/s (50 kbp ONT reads, viral index)", "value": R * world * a.steps / dt, "unit": "reads/s", "gbp_per_s": m.acc["bases"] * world / dt / 1e9,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64 integer", "data": "synthetic",
            "config": {"workload": "viral-RefSeq-sized index (BASELINE configs[1]; the RefSeq files are not in the mount): %s + %d distinct batches of %d synthetic %d bp ONT-15%%-error reads per GPU simulated from it" % (idx_desc, B, R, Lr),
                       "reads_per_gpu": R, "read_len": Lr, "batches_per_gpu": B, "parallelism": "reads sharded x%d, index replicated" % world},
            "index_build": build_info,
            "kernel_ms_per_step": m.kernel_ms(),
            "reads_in_early_launch": m.acc["early"] / steps, "reads_in_second_run": m.acc["retry"] / steps, "reads_handed_to_heavy_launch": m.acc["requeue"] / steps,
            "roofline": roof_cls if dom_is_cls else roof_seed,
            "roofline_seed_lookup": roof_seed,
            "end_to_end": e2e,
            "reads_mapped_frac": m.n_mapped / max(R, 1), "reads_with_device_status": m.n_bad,
            "per_read_wave_us": {"mean": sum(m.dev_us) / max(len(m.dev_us), 1), "median": m.dev_us[len(m.dev_us) // 2] if m.dev_us else 0,
                                 "p99": m.dev_us[int(len(m.dev_us) * 0.99)] if m.dev_us else 0, "max": m.dev_us[-1] if m.dev_us else 0},
            "generation_s": m.t_gen, "stage_s": t_stage,
        }
        lib_md5 = None; ns = min(a.cpu_sample, R)
        if not a.no_cpu_baseline or not a.no_cli:
            gpu_sam, fq_text = m.sample(ns)
            lib_md5 = hashlib.md5(gpu_sam).hexdigest()
        if not a.no_cpu_baseline:
            sample = os.path.join(SHM, "dsb_bench_sample.fq")
            with open(sample, "wb") as f:
                f.write(fq_text)
            out["cpu_baseline"], out["parity_sample"] = cpu_baseline(index_dir, sample, ns, ns * Lr, gpu_sam, a.t1_sample)
            os.remove(sample)
            # the seed lookup's numerator the survey's way: the probes the REFERENCE's scan makes (oracle-counted on the first reads of
            # the sample, scaled to the launch), beside the probes the kernel issued (device counters)
            oc = oracle_probe_counts(index_dir, fq_text)
            if oc:
                rs = out["roofline_seed_lookup"]; bases_launch = float(R) * Lr
                by = bases_launch * (1.0 + 64.0 * (oc["p0_per_base"] + oc["p1_per_base"]))
                issued = (rs["algorithmic_bytes"] - bases_launch) / 64.0 / bases_launch
                rs["oracle_numerator"] = dict(oc, algorithmic_bytes=by, achieved=by / (rs["ms"] / 1e3) / 1e9, frac=by / (rs["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
                                              issued_probes_per_base=issued, issued_over_oracle=issued / max(oc["p0_per_base"] + oc["p1_per_base"], 1e-9),
                                              what="L + 64 (P0 + P1) with P0, P1 counted by the CPU restatement of the reference's scan on the first %d reads of the sample" % oc["reads"])
            if e2e and out["cpu_baseline"]:
                e2e["vs_cpu_baseline"] = e2e["reads_per_s"] / world / out["cpu_baseline"]["value"]
    # ---- BASELINE configs[2]: 1 M synthetic 150 bp reads (1 % error) on the HEADLINE index, one resident batch
    if rank == 0 and not a.no_short_reads:
        out["config3_short_reads"] = resident_batch(D, L, idx, index_dir, local, 1 << 20, 150, 0.01, 4242, 0, gen_threads * world,
                                                    "1048576 synthetic 150 bp reads, 1 %% error, the headline index (%s), one batch resident in HBM" % ("viral-RefSeq-sized synthetic strain collection" if a.headline == "strain" else "demo"))
    m.close(); idx.close()
    # ---- BASELINE configs[4] proxy: a >= 1-Gbp index built here, PacBio-mixed reads, one resident batch + a parity sample against the reference
    if rank == 0 and not a.no_proxy:
        pdir, pinfo = proxy_index(a, D, local)
        pidx = D.Index(pdir)
        c5 = resident_batch(D, L, pidx, pdir, local, a.proxy_reads, 12000, 0.13, 3, 1, gen_threads * world,
                            "%d synthetic PacBio-mixed reads (log-normal lengths, mean 12 kbp, 13 %% error) on a %d-Mbp synthetic index built in this run (BASELINE configs[4] proxy), one batch resident in HBM" % (a.proxy_reads, a.proxy_mbp),
                            parity_reads=a.proxy_parity)
        c5["index_build"] = pinfo
        out["config5_proxy"] = c5
        pidx.close()
        shutil.rmtree(os.path.join(ROOT, "data", "bench_proxy"), ignore_errors=True)
    barrier()                                   # every rank has given its device memory back: the CLI takes all GPUs
    if rank == 0 and not a.no_cli:
        n_cli = a.cli_reads * world
        n_cli = max(R, min(n_cli, int(mem_limit() * 0.4 / rec_bytes)))
        out["cli_end_to_end"] = cli_end_to_end(index_dir, [Lr, 0.15, 1000, "ont", gen_threads * world], n_cli, list(range(world)) if not rehearsal else [0], lib_md5, ns)
        ce = out["cli_end_to_end"]
        if ce and out.get("cpu_baseline"):
            ce["vs_cpu_baseline"] = ce["reads_per_s"] / world / out["cpu_baseline"]["value"]
        if ce and e2e:
            ce["fraction_of_end_to_end"] = ce["reads_per_s"] / e2e["reads_per_s"]
    barrier()

    # ---- the reference's demo index (rank 0): the device path is 4.5x faster there, the host pipeline is what is measured --------
    if rank == 0 and not a.no_demo_index and world == 1:       # (secondary single-GPU measurements: reported by the N=1 run)
        idxd = D.Index(demo_dir)
        Bd = max(1, min(a.demo_batches, int(mem_limit() * 0.3 / (R * rec_bytes))))
        md = Measure(D, L, idxd, demo_dir, local, a, R, Lr, Bd, 1000, gen_threads * world)
        one = lambda v: v
        dtd = md.steps(1, a.demo_steps, torch.cuda.synchronize, one)
        rc, rs, _ = md.rooflines("demo")
        demo = {"workload": "demo viral-gs index (463 genomes, 11.5 Mbp, k=16 filter, 828 MB) + %d distinct batches of %d synthetic %d bp ONT-15%%-error reads, one GPU" % (Bd, R, Lr),
                "reads_per_s": R * a.demo_steps / dtd, "gbp_per_s": md.acc["bases"] / dtd / 1e9, "ms_per_step": dtd / a.demo_steps * 1e3, "steps": a.demo_steps,
                "kernel_ms_per_step": md.kernel_ms(), "roofline_k_classify": rc, "roofline_seed_lookup": rs,
                "reads_mapped_frac": md.n_mapped / max(R, 1), "reads_with_device_status": md.n_bad}
        if not a.no_end_to_end:
            demo["end_to_end"] = md.end_to_end(torch.cuda.synchronize, one, 1, min_batches=8)
        gpu_sam, fq_text = md.sample(ns)
        if not a.no_cpu_baseline:
            sample = os.path.join(SHM, "dsb_bench_sample.fq")
            with open(sample, "wb") as f:
                f.write(fq_text)
            demo["cpu_baseline"], demo["parity_sample"] = cpu_baseline(demo_dir, sample, ns, ns * Lr, gpu_sam, a.t1_sample)
            os.remove(sample)

        # ---- the seed-lookup kernel in the HBM regime: 2 x 2 GiB synthetic filter tables, 20 % full (SURVEY.md 8d asks for the roofline
        # claim on multi-GiB tables).  Only the seed lookup runs; answers on such tables are checked against a host recomputation in
        # tests/test_gpu_parity.py::test_seed_lookup_on_synthetic_multi_gib_tables.
        if not a.no_seed_hbm:
            ctx3 = D.Ctx(idxd, local, max_read_len=0, max_batch_reads=0, input_slots=1)
            ctx3.use_synthetic_filter(a.seed_hbm_mib << 20, 0.2)
            ctx3.upload_text(*md.bufs[0], R)
            ms = []; tm3 = None
            for _ in range(4):
                ctx3.run(); tm3 = ctx3.timing(); ms.append(tm3.seed_probe_ms)
            ms = sorted(ms[1:])[len(ms[1:]) // 2]
            by = tm3.bases + 64.0 * (tm3.windows + tm3.probes_t1)
            seed_hbm = {"kernel": "k_seed_scan" if tm3.seed_scan else "k_seed_probe", "bound": "hbm", "tables": "2 x %d MiB synthetic, 20 %% of the bits set (k = 18)" % a.seed_hbm_mib,
                        "ms": ms, "algorithmic_bytes": by, "achieved": by / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                        "probes_per_base": tm3.windows / max(tm3.bases, 1), "table1_probes_per_base": tm3.probes_t1 / max(tm3.bases, 1), "traffic": None}
            seed_hbm["traffic"], seed_hbm["traffic_source"] = pmc_traffic("r04_pmc_seed_hbm.json", seed_hbm["kernel"], reads_per_gpu=R, read_len=Lr, table_mib=a.seed_hbm_mib)
            ctx3.close()
            # the ceiling of the access pattern, measured in this run on this GPU: random 64-byte lines of a table as large as both filter tables
            ceil = gather_ceiling(2 * a.seed_hbm_mib)
            seed_hbm["gather_ceiling"] = {"GB/s": ceil, "what": "tools/gather_bench %d: one 4-byte word of a random 64-byte line per load, best of 4 / 8 / 16 loads in flight per lane" % (2 * a.seed_hbm_mib)}
            seed_hbm["frac_of_ceiling"] = seed_hbm["achieved"] / ceil if ceil else None
            out["roofline_seed_lookup_hbm"] = seed_hbm

        # ---- BASELINE configs[2] shape: 1 M synthetic 150 bp reads (1 % error) on the demo index, one resident batch
        if not a.no_short_reads:
            out["config2_short_reads"] = resident_batch(D, L, idxd, demo_dir, local, 1 << 20, 150, 0.01, 4242, 0, gen_threads * world,
                                                        "1048576 synthetic 150 bp reads, 1 % error, demo index, one batch resident in HBM")
        md.close(); idxd.close()
        if not a.no_cli:
            n_cli = max(R, min(a.demo_cli_reads, int(mem_limit() * 0.4 / rec_bytes)))
            demo["cli_end_to_end"] = ce = cli_end_to_end(demo_dir, [Lr, 0.15, 1000, "ont", gen_threads * world], n_cli, [0], hashlib.md5(gpu_sam).hexdigest(), ns)
            if ce and demo.get("cpu_baseline"):
                ce["vs_cpu_baseline"] = ce["reads_per_s"] / demo["cpu_baseline"]["value"]
            if ce and demo.get("end_to_end"):
                ce["fraction_of_end_to_end"] = ce["reads_per_s"] / demo["end_to_end"]["reads_per_s"]
        out["demo_index"] = demo
    if rank == 0:
        shutil.rmtree(os.path.join(ROOT, "data", "bench_strain"), ignore_errors=True)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
