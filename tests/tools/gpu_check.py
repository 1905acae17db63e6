"""GPU parity check: classify a FASTQ on the GPU and compare every hit with the CPU oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import desamba_amd as D
import oracle_lib as O

def main():
    index_dir = sys.argv[1]; fq = sys.argv[2]; limit = int(sys.argv[3]) if len(sys.argv) > 3 else None
    slots = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    t = time.time(); idx = D.Index(index_dir); print("index load %.2fs" % (time.time() - t), flush=True)
    recs = D.read_fastq(fq, limit); print("reads", len(recs), flush=True)
    t = time.time(); ctx = D.Ctx(idx, 0, n_slots=slots); print("ctx create (index -> HBM) %.2fs" % (time.time() - t), flush=True)
    reads = D.make_reads(recs)
    t = time.time(); ctx.upload(reads); print("upload %.2fs" % (time.time() - t), flush=True)
    t = time.time(); ctx.run(); print("run %.3fs" % (time.time() - t), flush=True)
    tm = ctx.timing(); print("timing ms: encode %.3f probe %.3f classify %.3f total %.3f windows %d p1 %d bases %d" % (tm.encode_ms, tm.seed_probe_ms, tm.classify_ms, tm.total_ms, tm.windows, tm.probes_t1, tm.bases), flush=True)
    res = ctx.fetch(strict=False)
    ora = O.Oracle(index_dir)
    bad = 0; badseed = 0; hist = 0; nst = 0
    for i, (name, seq, qual) in enumerate(recs):
        exp = ora.classify(seq, hist)
        hist = max(hist, len(seq))
        rr = res.reads[i]
        got = [res.hits[rr.first + k].key() for k in range(rr.n)]
        if rr.status: nst += 1
        if got != exp:
            bad += 1
            if bad <= 5:
                print("MISMATCH read", i, name, "len", len(seq), "status", rr.status, "\n  got", got[:3], "\n  exp", exp[:3], flush=True)
                for s in (1, 0):
                    gs = ctx.seeds(i, s); es = ora.seeds(s)
                    if gs != es: print("   seeds differ strand", s, len(gs[0]), len(es[0]), gs[1], es[1])
    print("reads %d mismatching %d status!=0 %d" % (len(recs), bad, nst), flush=True)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
