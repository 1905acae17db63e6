#!/bin/bash
# Collect rocprofv3 PMC counters for the bench workload, one pass per counter group (PMC only, no tracing).
#   tools/pmc_run.sh <tag> <reads> "<CTR CTR ...>" ["<CTR ...>" ...]      PMC_HEADLINE=demo: the demo index instead of the strain index built in the bench
# Writes gpurun_out/pmc_<tag>.json: per-kernel, per-launch averages of every counter.
# PMC_SEED_HBM_MIB=<MiB>: profile tools/seed_hbm_only.py (the seed lookup on synthetic tables of that size) instead of bench.py.
set -e -o pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
tag=$1; reads=$2; shift 2
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "$@"; do
	out=$ROOT/gpurun_out/pmc_${tag}_$i
	rm -rf "$out"
	if [ -n "$PMC_SEED_HBM_MIB" ]; then
		(cd /tmp && timeout -k 10 300 rocprofv3 --pmc $grp -d "$out" -o run --output-format csv -- python3 "$ROOT/tools/seed_hbm_only.py" "$reads" "$PMC_SEED_HBM_MIB" > "$out.log" 2>&1) || { echo "pass $i ($grp) failed"; tail -5 "$out.log"; }
	else
		(cd /tmp && timeout -k 10 300 rocprofv3 --pmc $grp -d "$out" -o run --output-format csv -- python3 "$ROOT/bench.py" --headline "${PMC_HEADLINE:-strain}" --no-cpu-baseline --no-end-to-end --no-cli --no-demo-index --no-proxy --no-short-reads --no-budget-build --steps 2 --warmup 0 --batches 1 --reads-per-gpu "$reads" > "$out.log" 2>&1) || { echo "pass $i ($grp) failed"; tail -5 "$out.log"; }
	fi
	i=$((i+1))
done
python3 - "$tag" "$reads" <<'PYEOF'
import csv, glob, json, sys, collections
sys.path.insert(0, ".")
import os
tag, reads = sys.argv[1], int(sys.argv[2])
mib = int(os.environ.get("PMC_SEED_HBM_MIB") or 0)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:                     # per launch: average over the dispatches of the kernel
    for c in acc[k]:
        acc[k][c] /= max(len(disp[k][c]), 1)
import desamba_amd as D
import bench
json.dump({"workload": {"reads_per_gpu": reads, "read_len": 50000, "library": D.lib().dsb_version().decode(), "device_source_md5": bench.device_source_md5(),
                        **({"table_mib": mib, "what": "%d x 50 kbp ONT reads, tools/seed_hbm_only.py on 2 x %d MiB synthetic tables; per-launch averages" % (reads, mib)} if mib else
                           {"index": os.environ.get("PMC_HEADLINE", "strain"), "what": "%d x 50 kbp ONT reads, bench.py --headline %s --steps 2 --warmup 0 --batches 1; per-launch averages" % (reads, os.environ.get("PMC_HEADLINE", "strain"))})}, "counters": acc},
          open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1, sort_keys=True)
for k in ("k_seed_scan", "k_classify"):
    print(k, json.dumps(acc.get(k, {}), sort_keys=True))
PYEOF
