#!/bin/bash
# A/B of two builds of the library on the strain index:  tests/tools/big_index_ab.sh <libA.so> <libB.so> [Mbp] [reads]
cd "$(dirname "$0")/../.."
A=$1; B=$2; MBP=${3:-380}; N=${4:-16384}; D=data/big; I=$D/index; mkdir -p gpurun_out $I
if [ ! -f $I/deSAMBA.ref_p ]; then
	python3 tools/synth_ref.py $D/syn.fa $MBP 1 2>&1
	oracle/_ref/kmer_srt $D/syn.fa $D/kmer.srt > gpurun_out/big_kmer.log 2>&1
	oracle/_ref/deSAMBA index $D/kmer.srt $D/syn.fa $I > gpurun_out/big_build.log 2>&1
	rm -f $D/kmer.srt; echo "index built"
fi
tools/readsim $I /dev/shm/y.fq $N 50000 0.15 1 ont > /dev/null 2>&1
tools/readsim $I /dev/shm/z.fq $((2 * N)) 12000 0.12 9 pacbio > /dev/null 2>&1
for lib in $A $B $A $B; do for f in y z; do
	echo "$lib $f: $(DSB_INDEX=$I DSB_LIB_PATH=$PWD/$lib timeout -k 10 200 python3 tools/prof_generic.py /dev/shm/$f.fq 2 2>&1 | tail -1)"
done; done
rm -f /dev/shm/y.fq /dev/shm/z.fq
