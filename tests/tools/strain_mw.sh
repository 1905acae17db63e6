#!/bin/bash
# The 380-Mbp synthetic strain index with short tandem repeats (tools/synth_ref.py defaults), 16384 x 50 kbp reads:
# device time with and without handing heavy reads (quadratic sparse DP) to workgroups of wavefronts, and the SAM
# against the reference's.
cd "$(dirname "$0")/../.."
D=data/big; I=$D/index; mkdir -p $D
if [ ! -f $I/deSAMBA.ref_p ]; then python3 tools/synth_ref.py $D/syn.fa 380 1 2>&1; desamba_amd/bin/deSAMBA index $D/syn.fa $I 2>&1 | tail -1; fi
tools/readsim $I /dev/shm/y.fq 16384 50000 0.15 1 ont > /dev/null 2>&1
for hp in "default 64" "default 256" "50000000 256" "default 512"; do
	set -- $hp; hp=$1; export DSB_HEAVY_MW=$2 DSB_HEAVY_FIRST=$2
	if [ $hp = default ]; then unset DSB_HEAVY_PREDS; else export DSB_HEAVY_PREDS=$hp; fi
	echo "DSB_HEAVY_PREDS=$hp early=MW=$DSB_HEAVY_MW: $(DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 3 2>&1 | tail -1)"
done
unset DSB_HEAVY_PREDS DSB_HEAVY_MW DSB_HEAVY_FIRST
$PWD/desamba_amd/bin/deSAMBA classify $I /dev/shm/y.fq -o /dev/shm/y_gpu.sam > /dev/null 2>&1
oracle/_ref/deSAMBA_ubfree classify -t $(tests/tools/host_cpus.sh) $I /dev/shm/y.fq -o /dev/shm/y_ref.sam > /dev/null 2>&1
cmp -s /dev/shm/y_gpu.sam /dev/shm/y_ref.sam && echo "SAM identical to the reference (default settings)" || echo "SAM DIFFERS"
rm -f /dev/shm/y.fq /dev/shm/y_gpu.sam /dev/shm/y_ref.sam
