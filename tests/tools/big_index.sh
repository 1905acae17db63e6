#!/bin/bash
# Index-size experiment on a GPU box: a synthetic reference collection of <Mbp> million bases
# (tools/synth_ref.py) is indexed by `deSAMBA index` of this repo (byte-identical to the reference's, tests/tools/big_build.sh), then the CLI of this repo runs against the reference's
# UB-pinned build on reads simulated from it, byte for byte, and the device path is timed kernel by kernel.
# >= 240 Mbp moves the exist tables to 2 x 256 MiB and the exist-k-mer length to 17 (src/idx.c:988-989,971).
#   tests/tools/big_index.sh [outdir] [Mbp]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; MBP=${2:-380}; D=data/big; mkdir -p "$OUT" $D/index
R=oracle/_ref/deSAMBA_ubfree; G=desamba_amd/bin/deSAMBA; I=$D/index; T=$(tests/tools/host_cpus.sh)
TIMEFORMAT="%R"
if [ ! -f $I/deSAMBA.ref_p ]; then
	python3 tools/synth_ref.py $D/syn.fa $MBP 1 2>&1
	t1=$( { time $G index $D/syn.fa $I > /dev/null 2> "$OUT/big_build.log"; } 2>&1 )
	echo "index built by this repo's builder in ${t1}s wall ($(tail -1 "$OUT/big_build.log")); $(du -sm $I | cut -f1) MB"
fi
for cfg in "ont50k 16384 50000 0.15 1 ont" "ngs150 500000 150 0.01 7 ngs" "pacbio 32768 12000 0.12 9 pacbio"; do
	set -- $cfg
	tools/readsim $I /dev/shm/y.fq $2 $3 $4 $5 $6 > /dev/null 2>&1
	tg=$( { time $G classify $I /dev/shm/y.fq -o /dev/shm/y_gpu.sam > /dev/null 2> "$OUT/big_$1_gpu.log"; } 2>&1 )
	tr=$( { time $R classify -t $T $I /dev/shm/y.fq -o /dev/shm/y_ref.sam > /dev/null 2> "$OUT/big_$1_ref.log"; } 2>&1 )
	if cmp -s /dev/shm/y_gpu.sam /dev/shm/y_ref.sam; then res="IDENTICAL ($(wc -l < /dev/shm/y_ref.sam) SAM lines)"; else res="DIFFER in $(diff /dev/shm/y_gpu.sam /dev/shm/y_ref.sam | grep -c '^<') lines"; diff /dev/shm/y_gpu.sam /dev/shm/y_ref.sam | head -4 | cut -c1-200; fi
	echo "$1 ($2 reads): $res; wall incl. index load: this CLI ${tg}s, reference -t $T ${tr}s; classify only: $(grep -ho 'processed in [0-9.]*s' "$OUT/big_$1_gpu.log") vs $(grep -ho 'processed in [0-9.]*s' "$OUT/big_$1_ref.log")"
	DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 3 2>&1 | tail -1
done
rm -f /dev/shm/y.fq /dev/shm/y_gpu.sam /dev/shm/y_ref.sam
