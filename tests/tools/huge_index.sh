#!/bin/bash
# BASELINE configs[4] proxy on a GPU box: a synthetic reference collection of <Mbp> million bases
# (default 3600: a BWT of > 2^32 rows, exist-k-mer tables of 2 x 2 GiB with k = 18, a raw 13-mer table) is indexed by
# `deSAMBA index` of this repo; <reads> PacBio-error reads of mixed lengths simulated from it are classified by the CLI; the
# first <sample> reads also by the reference's UB-pinned build on the same index directory, byte for byte.
#   tests/tools/huge_index.sh [outdir] [Mbp] [reads] [sample]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; MBP=${2:-3600}; N=${3:-65536}; S=${4:-2048}; D=data/huge; mkdir -p "$OUT" $D/index
R=oracle/_ref/deSAMBA_ubfree; G=desamba_amd/bin/deSAMBA; I=$D/index; T=$(python3 -c "import ctypes,os; print(ctypes.CDLL('desamba_amd/libdesamba_amd.so').dsb_host_cpus())")
TIMEFORMAT="%R"
t0=$( { time python3 tools/synth_ref.py $D/syn.fa $MBP 11 3 60 12 2> "$OUT/huge_synth.log"; } 2>&1 ); echo "reference: $(cat "$OUT/huge_synth.log") in ${t0}s, $(du -m $D/syn.fa | cut -f1) MB of FASTA"
t1=$( { time $G index $D/syn.fa $I > /dev/null 2> "$OUT/huge_build.log"; } 2>&1 )
echo "index built by this repo's builder in ${t1}s wall"; tail -2 "$OUT/huge_build.log"; ls -l $I | awk '{print $5, $9}'
rm -f $D/syn.fa
python3 tools/gen_fastq.py $I /dev/shm/h.fq $N 12000 0.13 3 pacbio 16
tg=$( { time DSB_CLI_TRACE=1 $G classify $I /dev/shm/h.fq -o /dev/shm/h_gpu.sam > /dev/null 2> "$OUT/huge_gpu.log"; } 2>&1 )
echo "CLI: ${tg}s wall incl. index load; $(grep -h 'processed in' "$OUT/huge_gpu.log")"; grep -h "trace\] input\|^\[gpu" "$OUT/huge_gpu.log" | cut -c1-200
python3 - /dev/shm/h.fq /dev/shm/hs.fq $S <<'PY'
import sys
n = int(sys.argv[3])
with open(sys.argv[1], "rb") as f, open(sys.argv[2], "wb") as g:
    for _ in range(4 * n):
        g.write(f.readline())
PY
tr=$( { time $R classify -t $T $I /dev/shm/hs.fq -o /dev/shm/hs_ref.sam > /dev/null 2> "$OUT/huge_ref.log"; } 2>&1 )
echo "reference (UB-pinned, -t $T) on the first $S reads: ${tr}s wall; $(grep -h 'processed in' "$OUT/huge_ref.log")"
python3 - /dev/shm/h_gpu.sam /dev/shm/hs_ref.sam $S <<'PY'
import sys
def by_read(path, limit=None):
    d = {}; order = []
    for ln in open(path, "rb"):
        k = ln.split(b"\t", 1)[0]
        if k not in d:
            if limit and len(order) >= limit: break
            d[k] = []; order.append(k)
        d[k].append(ln)
    return d, order
ref, ro = by_read(sys.argv[2]); gpu, go = by_read(sys.argv[1], len(ro))
bad = [k for k in ro if gpu.get(k) != ref[k]]
print("parity sample: %d reads, %d SAM lines of the reference; differing reads: %d; mapped reads in the sample: %d" % (len(ro), sum(len(v) for v in ref.values()), len(bad), sum(1 for k in ro if ref[k][0].split(b'\t')[1] != b'4')))
for k in bad[:3]:
    print(k, gpu.get(k), ref[k])
PY
rm -rf /dev/shm/h.fq /dev/shm/hs.fq /dev/shm/h_gpu.sam /dev/shm/hs_ref.sam $D
