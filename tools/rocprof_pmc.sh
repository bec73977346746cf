#!/bin/bash
# usage: tools/rocprof_pmc.sh <name> <counter> <python-script> [args...]
# One rocprofv3 --pmc pass (counters: one quoted, space separated argument; no trace domains) over `python3 <script> args`; leaves
# gpurun_out/<name>_<counter>.csv.  Run from the repo root on the GPU box.
set -u
name=$1; shift
counter=$1; shift
script=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$(echo "$counter" | tr " " "+")
out=$root/gpurun_out/pmc_${name}_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc $counter --output-format csv -d "$out" -o "$name" -- python3 "$root/$script" "$@" > "$root/gpurun_out/${name}_${tag}.log" 2>&1
rc=$?
cd "$root"
f=$(find "$out" -name "*counter_collection.csv" | sort | tail -1)
if [ -n "$f" ]; then cp "$f" "gpurun_out/${name}_${tag}.csv"; echo "ok $(wc -l < "$f") rows"; else echo "no counter csv (rc=$rc)"; tail -5 "gpurun_out/${name}_${tag}.log"; fi
exit $rc
