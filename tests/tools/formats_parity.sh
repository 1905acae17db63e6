#!/bin/bash
# GPU box: the CLI against the reference's UB-pinned build on three input files of very different reads in one run (200000 x 150 bp,
# 8192 x 50 kbp ONT, 8192 PacBio-mixed) in all four output formats, and with -l / -r / -s set.   tests/tools/formats_parity.sh [outdir]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.demo_dir()" > /dev/null 2>&1
I=data/demo/index; R=oracle/_ref/deSAMBA_ubfree; G=desamba_amd/bin/deSAMBA
tools/readsim $I /dev/shm/a.fq 8192 50000 0.15 41 ont > /dev/null 2>&1
tools/readsim $I /dev/shm/b.fq 200000 150 0.01 42 ngs > /dev/null 2>&1
tools/readsim $I /dev/shm/c.fq 8192 12000 0.12 43 pacbio > /dev/null 2>&1
for f in SAM SAM_FULL DES DES_FULL; do
	$G classify -f $f $I /dev/shm/b.fq /dev/shm/a.fq /dev/shm/c.fq -o /dev/shm/g.out > /dev/null 2> "$OUT/fmt_g.log"
	$R classify -t $(tests/tools/host_cpus.sh) -f $f $I /dev/shm/b.fq /dev/shm/a.fq /dev/shm/c.fq -o /dev/shm/r.out > /dev/null 2>&1
	if cmp -s /dev/shm/g.out /dev/shm/r.out; then echo "$f, three files (short, ONT 50k, PacBio): IDENTICAL ($(wc -l < /dev/shm/r.out) lines), $(grep -ho 'processed in [0-9.]*s' "$OUT/fmt_g.log")"
	else echo "$f: DIFFER"; diff /dev/shm/g.out /dev/shm/r.out | head -4 | cut -c1-200; fi
done
$G classify -l 500 -r 3 -s 80 $I /dev/shm/a.fq -o /dev/shm/g.out > /dev/null 2>&1
$R classify -t $(tests/tools/host_cpus.sh) -l 500 -r 3 -s 80 $I /dev/shm/a.fq -o /dev/shm/r.out > /dev/null 2>&1
cmp -s /dev/shm/g.out /dev/shm/r.out && echo "-l 500 -r 3 -s 80: IDENTICAL" || echo "-l 500 -r 3 -s 80: DIFFER"
rm -f /dev/shm/a.fq /dev/shm/b.fq /dev/shm/c.fq /dev/shm/g.out /dev/shm/r.out
