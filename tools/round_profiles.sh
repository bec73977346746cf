#!/bin/bash
# Round-end measurement on the GPU box (run from the repo root): GPU tests, the default bench line, rocprofv3
# kernel-trace stats of the bench command and of the extractor chain alone, and the two PMC passes behind
# profiles/*_hbm_traffic_pmc.json -- taken AT THE BENCH'S OWN LAUNCH SIZE (default: 512 stereo frames = 1024 images per
# extractor launch, 3 BA partitions of 171 windows). Everything lands in gpurun_out/; copy what should be judged into profiles/.
set -u
tag=${1:-r02}
frames=${2:-512}
split=${3:-3}
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/${tag}_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json.log 2>gpurun_out/${tag}_bench.err; echo "bench rc=$?"
tools/rocprof_stats.sh ${tag}_bench bench.py --no-cpu-baseline > /dev/null; echo "stats bench rc=$?"
tools/rocprof_stats.sh ${tag}_extract tools/prof_extract.py $frames 3 64 > /dev/null; echo "stats extract rc=$?"
tools/rocprof_pmc.sh ${tag} FETCH_SIZE bench.py --steps 2 --warmup 1 --frames $frames --ba-split $split --no-cpu-baseline --distinct 16 --ba-distinct 8
tools/rocprof_pmc.sh ${tag} WRITE_SIZE bench.py --steps 2 --warmup 1 --frames $frames --ba-split $split --no-cpu-baseline --distinct 16 --ba-distinct 8
python tools/pmc_traffic.py gpurun_out/${tag}_FETCH_SIZE.csv gpurun_out/${tag}_WRITE_SIZE.csv gpurun_out/${tag}_hbm_traffic_pmc.json \
  "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over: python3 bench.py --steps 2 --warmup 1 --frames $frames --ba-split $split --no-cpu-baseline --distinct 16 --ba-distinct 8 (the bench's own launch size, 1280x720); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction in MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced reads; uncalibrated for narrow gathers)" \
  "images=$((2 * frames)),windows=$(( frames / split )),pairs=$frames"
# BASELINE configs[3]'s per-GPU share: 8 stereo frames per step, one BA partition
timeout -k 10 300 python bench.py --frames 8 --steps 50 --no-cpu-baseline > gpurun_out/${tag}_bench_f8.json.log 2>gpurun_out/${tag}_bench_f8.err; echo "bench f8 rc=$?"
tools/rocprof_stats.sh ${tag}_f8 bench.py --frames 8 --steps 50 --no-cpu-baseline > /dev/null; echo "stats f8 rc=$?"
cp gpurun_out/prof_${tag}_f8/*kernel_trace.csv gpurun_out/${tag}_f8_kernel_trace.csv 2>/dev/null
# the BA chain alone (171 windows) and the matrix-instruction count of its Schur kernel
timeout -k 10 300 python tools/prof_ba.py 171 2 16 > gpurun_out/${tag}_prof_ba_171.log 2>&1; echo "prof_ba rc=$?"
tools/rocprof_pmc.sh ${tag}_ba "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVES" tools/prof_ba.py 171 1 16
cat gpurun_out/${tag}_bench.json.log
