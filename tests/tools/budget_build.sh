#!/bin/bash
# The index builder beyond the device's memory at scale on a GPU box: a synthetic collection of <Mbp>
# million bases is indexed in one piece and under DSB_BUILD_BUDGET=<budget> (ranges of k-mer prefixes, dsb_build_parts.h); the ten files are
# compared byte for byte, stage times and what each stage held are printed.
#   tests/tools/budget_build.sh [outdir] [Mbp] [budget, e.g. 20g]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out}; MBP=${2:-3600}; BUD=${3:-20g}; D=data/budget; mkdir -p "$OUT" $D/a $D/b
G=desamba_amd/bin/deSAMBA
TIMEFORMAT="%R"
t0=$( { time python3 tools/synth_ref.py $D/syn.fa $MBP 11 3 60 12 2> "$OUT/budget_synth.log"; } 2>&1 ); echo "reference: $(cat "$OUT/budget_synth.log") in ${t0}s"
t1=$( { time $G index $D/syn.fa $D/a > /dev/null 2> "$OUT/budget_a.log"; } 2>&1 )
echo "one piece: ${t1}s wall"; tail -2 "$OUT/budget_a.log"
t2=$( { time DSB_BUILD_TRACE=1 DSB_BUILD_BUDGET=$BUD $G index $D/syn.fa $D/b > /dev/null 2> "$OUT/budget_b.log"; } 2>&1 )
echo "budget $BUD: ${t2}s wall"; tail -4 "$OUT/budget_b.log"
bad=0
for f in $(ls $D/a); do if cmp -s $D/a/$f $D/b/$f; then :; else echo "DIFFERS: $f"; bad=1; fi; done
[ $bad = 0 ] && echo "all $(ls $D/a | wc -l) files identical ($(du -sm $D/a | cut -f1) MB)"
grep -E "VmHWM" /proc/self/status > /dev/null; rm -rf $D
