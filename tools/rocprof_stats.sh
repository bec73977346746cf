#!/bin/bash
# usage: tools/rocprof_stats.sh <name> <python-script> [args...]
# Runs `rocprofv3 --kernel-trace --stats` over `python3 <script> args` (the program itself after `--`, no exec hops)
# and leaves gpurun_out/<name>_kernel_stats.csv.  Run from the repo root on the GPU box.
set -u
name=$1; shift
script=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$name" -- python3 "$root/$script" "$@" > "$root/gpurun_out/${name}.log" 2>&1
rc=$?
cd "$root"
f=$(find "$out" -name "*kernel_stats.csv" | sort | tail -1)
if [ -n "$f" ]; then
    cp "$f" "gpurun_out/${name}_kernel_stats.csv"
    cut -d, -f1-7 "gpurun_out/${name}_kernel_stats.csv" | cut -c1-160
else
    echo "no kernel_stats.csv produced (rc=$rc)"; tail -5 "gpurun_out/${name}.log"
fi
exit $rc
