#!/bin/bash
# The CLI's end-to-end rate on a GPU box: `deSAMBA classify` on N x 50 kbp reads in /dev/shm with its
# trace summary (reader / writer GB/s, per-worker busy fraction), the host I/O microbenchmark behind the reader's
# design, and a parity check of the CLI against the reference's UB-pinned build on a sample.
#   tests/tools/cli_rate.sh [outdir] [n_reads]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; N=${2:-262144}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.demo_dir()" > "$OUT/demo.log" 2>&1
R=oracle/_ref/deSAMBA_ubfree; G=desamba_amd/bin/deSAMBA; I=data/demo/index; T=$(tests/tools/host_cpus.sh)
gcc -O2 -o /tmp/iobench tools/iobench.c -lpthread
python tools/gen_fastq.py $I /dev/shm/s.fq 8192 50000 0.15 1001 ont 16
echo "== host I/O (0.8 GB file, then a 6.6 GB one)"
/tmp/iobench /dev/shm/s.fq 16
python tools/gen_fastq.py $I /dev/shm/a.fq 65536 50000 0.15 1001 ont 16
for t in 16; do /tmp/iobench /dev/shm/a.fq $t; done
echo "== parity sample: CLI vs reference (UB-pinned) on 8192 reads"
$G classify $I /dev/shm/s.fq -o /dev/shm/s_gpu.sam 2> "$OUT/cli_rate_s.log"
$R classify -t $T $I /dev/shm/s.fq -o /dev/shm/s_ref.sam > /dev/null 2>&1
cmp /dev/shm/s_gpu.sam /dev/shm/s_ref.sam && echo "IDENTICAL ($(wc -l < /dev/shm/s_ref.sam) lines)"
echo "== CLI on 65536 reads"
DSB_CLI_TRACE=1 $G classify $I /dev/shm/a.fq -o /dev/shm/a.sam 2> "$OUT/cli_rate_a.log"; grep -E "processed|trace|CPU" "$OUT/cli_rate_a.log"
md5sum /dev/shm/a.sam
echo "== .gz input (BGZF, written by tools/bgzip_lite.c): the CLI on 65536 reads, the reference on the first 8192 of them"
gcc -O2 -o /tmp/bgzip_lite tools/bgzip_lite.c -lz -lpthread
/tmp/bgzip_lite /dev/shm/a.fq /dev/shm/a.fq.gz 16 1; /tmp/bgzip_lite /dev/shm/s.fq /dev/shm/s.fq.gz 16 1; ls -la /dev/shm/a.fq.gz /dev/shm/s.fq.gz
DSB_CLI_TRACE=1 $G classify $I /dev/shm/a.fq.gz -o /dev/shm/a_gz.sam 2> "$OUT/cli_rate_agz.log"; grep -E "processed|trace|CPU" "$OUT/cli_rate_agz.log"
md5sum /dev/shm/a_gz.sam
$R classify -t $T $I /dev/shm/s.fq.gz -o /dev/shm/s_ref_gz.sam 2>&1 | grep processed
$R classify -t $T $I /dev/shm/s.fq -o /dev/shm/s_ref.sam 2>&1 | grep processed
echo "== plain single-member gzip of the 8192 reads (one serial zlib stream, inflated ahead): CLI, then reference"
gzip -1 -c /dev/shm/s.fq > /dev/shm/s1.fq.gz
DSB_CLI_TRACE=1 $G classify $I /dev/shm/s1.fq.gz -o /dev/shm/s1.sam 2>&1 | grep -E "processed|inflate"
$R classify -t $T $I /dev/shm/s1.fq.gz -o /dev/shm/s1_ref.sam 2>&1 | grep processed
cmp /dev/shm/s1.sam /dev/shm/s_ref.sam && echo "gzip input: SAM identical"
rm -f /dev/shm/a.fq /dev/shm/a.sam /dev/shm/a.fq.gz /dev/shm/a_gz.sam /dev/shm/s1* /dev/shm/s.fq.gz /dev/shm/s_ref_gz.sam
python tools/gen_fastq.py $I /dev/shm/big.fq $N 50000 0.15 1001 ont 16
echo "== CLI on $N reads"
for rep in 1 2; do
	DSB_UPLOAD_TRACE=1 DSB_CLI_TRACE=1 $G classify $I /dev/shm/big.fq -o /dev/shm/big.sam 2> "$OUT/cli_rate_big$rep.log"; grep -E "processed|trace|CPU|upload\]" "$OUT/cli_rate_big$rep.log"
done
md5sum /dev/shm/big.sam
rm -f /dev/shm/big.fq /dev/shm/big.sam /dev/shm/s.fq /dev/shm/s_gpu.sam /dev/shm/s_ref.sam
