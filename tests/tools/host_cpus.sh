#!/bin/bash
# CPUs the reference may really use: the affinity mask capped by the control group's CPU quota (nproc ignores the quota: on the
# GPU box it says 256 where 16 are allowed, and a reference run with -t 256 is throttled -- ADVICE r03)
n=$(nproc); q=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)
set -- $q
if [ -n "$1" ] && [ "$1" != max ] && [ -n "$2" ]; then c=$(( ($1 + $2 - 1) / $2 )); [ $c -lt $n ] && n=$c; fi
[ $n -lt 1 ] && n=1
echo $n
