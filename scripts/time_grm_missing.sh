#!/bin/bash
# GRM stage time of the BASELINE configs[2] shape at several missing-call rates, sparse correction (k_grm_miss.hip) against
# the fp16 split kernel.  GPU box; run through gpurun from the repo root.
for rate in 0.0005 0.001 0.003 0.01; do
  for mode in 1 0; do
    JXGPU_GRM_MISS=$mode JXGPU_GRM_MISS_MAX=1 JXGPU_GRM_MISS_DENSE_MIN=${DENSE_MIN:-1} python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --missing $rate 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('missing $rate correction=$mode grm_ms', round(s['grm'],1), 'step_ms', round(d['ms_per_step'],1))"
  done
done
