#!/bin/bash
# Full-size drop-in check on a GPU box: the CLI of this repo against the reference's UB-pinned build
# (oracle/_ref/deSAMBA_ubfree -t <cores>) on four synthetic workloads, byte for byte, with wall times.
#   tests/tools/full_parity.sh [outdir]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.demo_dir()" > "$OUT/demo.log" 2>&1
R=oracle/_ref/deSAMBA_ubfree; G=desamba_amd/bin/deSAMBA; I=data/demo/index; T=$(tests/tools/host_cpus.sh)
TIMEFORMAT="%R"
SEED_SHIFT=${DSB_PARITY_SEED_SHIFT:-0}                 # other read sets: DSB_PARITY_SEED_SHIFT=100 tests/tools/full_parity.sh
for cfg in "ont50k 65536 50000 0.15 1 ont" "ngs150 1000000 150 0.01 7 ngs" "pacbio 65536 12000 0.12 9 pacbio" "ont8k_e25 20000 8000 0.25 5 ont"; do
	set -- $cfg
	tools/readsim $I /dev/shm/x.fq $2 $3 $4 $(($5 + SEED_SHIFT)) $6 > /dev/null 2>&1
	tg=$( { time $G classify $I /dev/shm/x.fq -o /dev/shm/x_gpu.sam > /dev/null 2> "$OUT/full_$1_gpu.log"; } 2>&1 )
	tr=$( { time $R classify -t $T $I /dev/shm/x.fq -o /dev/shm/x_ref.sam > /dev/null 2> "$OUT/full_$1_ref.log"; } 2>&1 )
	if cmp -s /dev/shm/x_gpu.sam /dev/shm/x_ref.sam; then res="IDENTICAL ($(wc -l < /dev/shm/x_ref.sam) SAM lines)"; else res="DIFFER in $(diff /dev/shm/x_gpu.sam /dev/shm/x_ref.sam | grep -c '^<') lines"; diff /dev/shm/x_gpu.sam /dev/shm/x_ref.sam | head -4 | cut -c1-200; fi
	echo "$1 ($2 reads): $res; wall incl. index load: this CLI ${tg}s, reference -t $T ${tr}s; classify only: $(grep -ho 'processed in [0-9.]*s' "$OUT/full_$1_gpu.log") vs $(grep -ho 'processed in [0-9.]*s' "$OUT/full_$1_ref.log")"
done
rm -f /dev/shm/x.fq /dev/shm/x_gpu.sam /dev/shm/x_ref.sam
