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
cat gpurun_out/${tag}_bench.json.log
