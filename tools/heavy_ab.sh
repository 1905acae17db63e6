#!/bin/bash
# The tandem-repeat-rich 380-Mbp index (tests/tools/big_index.sh) with different numbers of reads on eight wavefronts from the start
# (DSB_HEAVY_MW): tools/heavy_ab.sh [outdir] "<values...>"
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out}; D=data/big; mkdir -p "$OUT" $D/index; I=$D/index
if [ ! -f $I/deSAMBA.ref_p ]; then python3 tools/synth_ref.py $D/syn.fa 380 1 2>&1; desamba_amd/bin/deSAMBA index $D/syn.fa $I > /dev/null 2>&1; rm -f $D/syn.fa; fi
tools/readsim $I /dev/shm/y.fq 16384 50000 0.15 1 ont > /dev/null 2>&1
tools/readsim $I /dev/shm/z.fq 32768 12000 0.12 9 pacbio > /dev/null 2>&1
for v in ${2:-default 64 256}; do
	for f in y z; do
		if [ $v = default ]; then r=$(DSB_INDEX=$I timeout -k 10 200 python3 tools/prof_generic.py /dev/shm/$f.fq 3 2>&1 | tail -1); else r=$(DSB_HEAVY_MW=$v DSB_INDEX=$I timeout -k 10 200 python3 tools/prof_generic.py /dev/shm/$f.fq 3 2>&1 | tail -1); fi
		echo "DSB_HEAVY_MW=$v $f: $r"
	done
done
rm -f /dev/shm/y.fq /dev/shm/z.fq
