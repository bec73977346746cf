#!/bin/bash
# Round-end measurement on the GPU box (run from the repo root): GPU tests, the default bench line, rocprofv3
# kernel-trace stats of the bench command and of the two chains alone, and the two PMC passes behind
# profiles/*_hbm_traffic_pmc.json.  Everything lands in gpurun_out/; copy what should be judged into profiles/.
set -u
tag=${1:-r01}
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/${tag}_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json.log 2>gpurun_out/${tag}_bench.err; echo "bench rc=$?"
tools/rocprof_stats.sh ${tag}_bench bench.py --no-cpu-baseline > /dev/null; echo "stats bench rc=$?"
tools/rocprof_stats.sh ${tag}_extract tools/prof_extract.py 64 3 > /dev/null; echo "stats extract rc=$?"
tools/rocprof_stats.sh ${tag}_ba tools/prof_ba.py 64 3 > /dev/null; echo "stats ba rc=$?"
tools/rocprof_pmc.sh ${tag} FETCH_SIZE bench.py --steps 2 --warmup 1 --frames 64 --ba-split 1 --no-cpu-baseline
tools/rocprof_pmc.sh ${tag} WRITE_SIZE bench.py --steps 2 --warmup 1 --frames 64 --ba-split 1 --no-cpu-baseline
python tools/pmc_traffic.py gpurun_out/${tag}_FETCH_SIZE.csv gpurun_out/${tag}_WRITE_SIZE.csv gpurun_out/${tag}_hbm_traffic_pmc.json \
  "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over: python3 bench.py --steps 2 --warmup 1 --frames 64 --ba-split 1 --no-cpu-baseline (64 stereo frames / 64 BA windows per launch, 1280x720); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction in MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced reads; uncalibrated for narrow gathers)"
cat gpurun_out/${tag}_bench.json.log
