#!/bin/bash
# Full-batch drop-in check on the HEADLINE index (the 321-Mbp synthetic strain collection of bench.py) on a GPU box: <reads> fresh 50-kbp ONT reads
# through this repo's CLI and through the reference's UB-pinned build on the same index directory, byte for byte.   tests/tools/headline_parity.sh [outdir] [reads] [seed]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; N=${2:-32768}; SEED=${3:-4242}; D=data/headline; mkdir -p "$OUT" $D/index
R=oracle/_ref/deSAMBA_ubfree; G=desamba_amd/bin/deSAMBA; I=$D/index; T=$(tests/tools/host_cpus.sh)
TIMEFORMAT="%R"
python3 tools/synth_ref.py $D/syn.fa 320 11 3 60 12 2>&1
$G index $D/syn.fa $I > /dev/null 2> "$OUT/headline_build.log"; tail -1 "$OUT/headline_build.log"; rm -f $D/syn.fa
tools/readsim $I /dev/shm/hl.fq $N 50000 0.15 $SEED ont > /dev/null 2>&1
tg=$( { time $G classify $I /dev/shm/hl.fq -o /dev/shm/hl_gpu.sam > /dev/null 2> "$OUT/headline_gpu.log"; } 2>&1 )
tr=$( { time $R classify -t $T $I /dev/shm/hl.fq -o /dev/shm/hl_ref.sam > /dev/null 2> "$OUT/headline_ref.log"; } 2>&1 )
if cmp -s /dev/shm/hl_gpu.sam /dev/shm/hl_ref.sam; then res="IDENTICAL ($(wc -l < /dev/shm/hl_ref.sam) SAM lines)"; else res="DIFFER in $(diff /dev/shm/hl_gpu.sam /dev/shm/hl_ref.sam | grep -c '^<') lines"; fi
echo "headline index, $N reads x 50 kbp (seed $SEED): $res; classify only: $(grep -ho 'processed in [0-9.]*s' "$OUT/headline_gpu.log") vs $(grep -ho 'processed in [0-9.]*s' "$OUT/headline_ref.log") (reference -t $T)"
rm -rf /dev/shm/hl.fq /dev/shm/hl_gpu.sam /dev/shm/hl_ref.sam $D
