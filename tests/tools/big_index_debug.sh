#!/bin/bash
# Companion of big_index.sh: where the time of the 50-kbp workload goes on the synthetic strain index
# (per-read wave times, then the DSB_DEBUG stage split).   tests/tools/big_index_debug.sh [outdir] [Mbp] [reads]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; MBP=${2:-380}; N=${3:-16384}; D=data/big; I=$D/index; mkdir -p "$OUT" $I
if [ ! -f $I/deSAMBA.ref_p ]; then
	python3 tools/synth_ref.py $D/syn.fa $MBP 1 2>&1
	oracle/_ref/kmer_srt $D/syn.fa $D/kmer.srt > "$OUT/big_kmer.log" 2>&1
	oracle/_ref/deSAMBA index $D/kmer.srt $D/syn.fa $I > "$OUT/big_build.log" 2>&1
	rm -f $D/kmer.srt; echo "index built"
fi
tools/readsim $I /dev/shm/y.fq $N 50000 0.15 1 ont > /dev/null 2>&1
DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 2 2>&1 | tail -1
DSB_INDEX=$I DSB_DEBUG=1 timeout -k 10 300 python3 tools/prof_generic.py /dev/shm/y.fq 1 2>&1 | grep -v "still running\|   slot" | tail -8
rm -f /dev/shm/y.fq
