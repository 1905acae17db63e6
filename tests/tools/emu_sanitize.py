import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import __graft_entry__ as G
import desamba_amd as D, emu_lib
d = G.demo_dir(); idx = os.path.join(d,'index')
e = emu_lib.Emu(idx)
GOLDEN=os.path.join(ROOT,'tests','golden')
n=0; hist=0
def run(recs, tag):
    global n, hist
    for name, seq, q in recs:
        e.classify(seq, hist); hist=max(hist,len(seq)); n+=1
    print(tag, 'ok', n, flush=True)
run(D.read_fastq(os.path.join(d,'ERR1050068.fastq'), 200), 'demo')
for nm in ["ont20k", "ngs150", "pb", "ont5k_e25", "appc", "heavy", "wrapq", "ngs_e14", "overhang", "manyanchors"]:
    run(D.read_fastq(os.path.join(GOLDEN,'synth',nm+'.fq')), nm)
run([(b"short", b"ACGT" * 9, None), (b"min", b"ACGTTGCA" * 5, None), (b"polyA", b"A" * 300, None), (b"allN", b"N" * 200, None), (b"lower", b"acgtnnacgt" * 30, None), (b"l39", b"A" * 39, None), (b"empty", b"", None)], 'edge')
fq = os.path.join(os.environ.get("TMPDIR", "/tmp"), "dsb_emu_sanitize.fq")
subprocess.check_call([os.path.join(ROOT,'tools','readsim'), idx, fq, '150', '50000', '0.15', '1', 'ont'])
run(D.read_fastq(fq), 'ont50k')
