#!/bin/bash
# Where the time goes on the 380-Mbp synthetic strain index (tools/synth_ref.py defaults: with short tandem repeats):
# stage split of k_classify over the batch and of its slowest read (DSB_DEBUG=1).
#   tests/tools/strain_tail.sh [Mbp] [reads] [read_len] [err] [profile]
cd "$(dirname "$0")/../.."
MBP=${1:-380}; N=${2:-16384}; LEN=${3:-50000}; ERR=${4:-0.15}; PROF=${5:-ont}
D=data/big; I=$D/index; mkdir -p $D
if [ ! -f $I/deSAMBA.ref_p ]; then
	python3 tools/synth_ref.py $D/syn.fa $MBP 1 2>&1
	desamba_amd/bin/deSAMBA index $D/syn.fa $I 2>&1 | tail -1
fi
tools/readsim $I /dev/shm/y.fq $N $LEN $ERR 1 $PROF > /dev/null 2>&1
DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 3 2>&1 | tail -1
DSB_DEBUG=1 DSB_INDEX=$I python3 tools/prof_generic.py /dev/shm/y.fq 2 2>&1 | grep -v "^\[dsb\] encode\|slot " | tail -5
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import desamba_amd as D
idx = D.Index("data/big/index"); ctx = D.Ctx(idx, 0)
n = ctx.upload_fastq("/dev/shm/y.fq"); ctx.run(); ctx.run()
res = ctx.fetch(strict=False); t = ctx.timing()
us = sorted(((res.reads[i].device_us, i, res.reads[i].n_anc) for i in range(n)), reverse=True)
print("ms: seed %.1f classify %.1f tail %.1f total %.1f  early %d mw %d" % (t.seed_probe_ms, t.classify_ms, t.tail_ms, t.total_ms, t.n_early, t.n_heavy_mw))
print("slowest reads (ms, index, anchors):", [(round(u / 1e3, 1), i, a) for u, i, a in us[:16]])
tot = sum(u for u, _, _ in us)
print("wave-time: total %.1f s, mean %.2f ms, median %.2f ms, p99 %.2f ms; top 16 reads hold %.1f %%, top 256 %.1f %%" % (tot / 1e6, tot / n / 1e3, us[n // 2][0] / 1e3, us[n // 100][0] / 1e3,
      100.0 * sum(u for u, _, _ in us[:16]) / tot, 100.0 * sum(u for u, _, _ in us[:256]) / tot))
PY
rm -f /dev/shm/y.fq
