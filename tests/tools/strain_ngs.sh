#!/bin/bash
# 500000 x 150 bp reads on the 380-Mbp synthetic strain index: a read per lane for the anchor stage (group mode) vs a read per wavefront
cd "$(dirname "$0")/../.."
D=data/big; I=$D/index; mkdir -p $D
if [ ! -f $I/deSAMBA.ref_p ]; then python3 tools/synth_ref.py $D/syn.fa 380 1 2>&1; desamba_amd/bin/deSAMBA index $D/syn.fa $I 2>&1 | tail -1; fi
tools/readsim $I /dev/shm/y.fq 500000 150 0.01 7 ngs > /dev/null 2>&1
for t in 1 24576 98304; do echo "group mode, first $t reads singly: $(DSB_GROUP_HEAD=$t DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 3 2>&1 | tail -1)"; done
echo "read per wave: $(DSB_NO_GROUP=1 DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 3 2>&1 | tail -1)"
rm -f /dev/shm/y.fq
