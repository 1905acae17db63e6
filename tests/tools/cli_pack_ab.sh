#!/bin/bash
# A/B on one GPU box: the CLI with the sequences packed to 2 bits per base by the gather threads (default) against DSB_UPLOAD_TEXT=1
# (round 3's text gather), demo index, N x 50 kbp reads in /dev/shm; the SAM of both runs must be the same.
#   tests/tools/cli_pack_ab.sh [outdir] [n_reads]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; N=${2:-262144}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.demo_dir()" > "$OUT/demo.log" 2>&1
G=desamba_amd/bin/deSAMBA; I=data/demo/index
python tools/gen_fastq.py $I /dev/shm/big.fq $N 50000 0.15 1001 ont 16
for rep in 1 2; do
	for mode in packed text; do
		if [ $mode = text ]; then export DSB_UPLOAD_TEXT=1; else unset DSB_UPLOAD_TEXT; fi
		DSB_CLI_TRACE=1 $G classify $I /dev/shm/big.fq -o /dev/shm/big_$mode.sam 2> "$OUT/cli_pack_$mode$rep.log"
		echo "$mode: $(grep -h 'processed in' "$OUT/cli_pack_$mode$rep.log") | $(grep -h 'worker 0' "$OUT/cli_pack_$mode$rep.log" | sed -E 's/.*upload ([0-9.]+ s \([^)]*\)).*/upload \1/')"
	done
done
unset DSB_UPLOAD_TEXT
cmp /dev/shm/big_packed.sam /dev/shm/big_text.sam && echo "SAM identical ($(wc -l < /dev/shm/big_text.sam) lines)"
rm -f /dev/shm/big.fq /dev/shm/big_packed.sam /dev/shm/big_text.sam
