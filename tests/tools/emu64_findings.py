"""The device code as 64 lanes with a race detector (tests/emu/emu_simt.cpp) over read sets: hits against the oracle, and what the
detector found, with source lines.   python3 tests/tools/emu64_findings.py [demo:N] [synth names...]   (CPU only)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
LIB = os.path.join(ROOT, "tests", "emu", "libdsbemu64.so")
os.environ["DSB_EMU_LIB"] = LIB
import __graft_entry__ as G
import desamba_amd as D, emu_lib, oracle_lib

def sym(off):
    out = subprocess.run(["addr2line", "-f", "-C", "-i", "-e", LIB, off], stdout=subprocess.PIPE).stdout.decode().split("\n")
    places = [out[i + 1].replace(ROOT + "/", "") for i in range(0, len(out) - 1, 2)]
    return " <- ".join(p.split("/")[-1] for p in places[:3])

def main():
    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu64.sh")])
    d = G.demo_dir(); idx = os.path.join(d, "index")
    e = emu_lib.Emu(idx); o = oracle_lib.Oracle(idx)
    sets = sys.argv[1:] or ["demo:60"]
    found = {}; bad = 0; n = 0; t0 = time.time()
    for s in sets:
        recs = D.read_fastq(os.path.join(d, "ERR1050068.fastq"), int(s.split(":")[1])) if s.startswith("demo:") else D.read_fastq(os.path.join(ROOT, "tests", "golden", "synth", s + ".fq"))
        hist = 0
        for name, seq, q in recs:
            exp = o.classify(seq, hist); got = e.classify(seq, hist); n += 1
            if got != exp:
                bad += 1; print("DIFFERENT HITS", s, name.decode(), len(seq))
            for f in e.findings():
                kind = f.split(" x")[0]; a, b = f.split(" at ")[1].split(); reg = f.split(" in ")[1].split(" at ")[0]
                key = (kind, a, b)
                if key not in found:
                    found[key] = [0, reg, "%s %s" % (s, name.decode())]
                found[key][0] += int(f.split(" x")[1].split()[0])
            hist = max(hist, len(seq))
    print("%d reads in %.1f s, %d with hits different from the oracle's, %d distinct findings" % (n, time.time() - t0, bad, len(found)))
    for (kind, a, b), (cnt, reg, first) in sorted(found.items(), key=lambda kv: -kv[1][0]):
        print("%-9s x%-7d %-28s %s   |   %s   (first: %s)" % (kind, cnt, reg, sym(a), sym(b) if b != "0" else "", first))
    return 1 if bad or found else 0

if __name__ == "__main__":
    sys.exit(main())
