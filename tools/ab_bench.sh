#!/bin/bash
# usage: tools/ab_bench.sh <variant-name|base> ...   -- default bench (no CPU leg) per experiment build, prints frames/s and ms per step
for v in "$@"; do
  if [ "$v" = base ]; then unset TB_HIP_LIB; else export TB_HIP_LIB=$(pwd)/build/variants/libtb_$v.so; fi
  python bench.py --no-cpu-baseline $AB_ARGS > gpurun_out/ab_$v.log 2>&1
  echo "$v: $(python -c "import json;d=json.loads(open('gpurun_out/ab_$v.log').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],'extract',d['roofline']['extractor']['isolated_chain_ms_per_step'], 'schur iso', d['roofline']['isolated_kernels_ms_per_step']['k_ba_schur'])")"
done
