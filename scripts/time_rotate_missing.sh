#!/bin/bash
# Rotation stage time of the BASELINE configs[2] shape at low missing-call rates: gather correction behind the int8 rotation
# (jxg_rotate_missing_correct) against the fp16 kernel (JXGPU_ROT_MISS_MAX=0).  GPU box; run through gpurun from the repo root.
for rate in 0.0005 0.001 0.002 0.003; do
  for mm in default 0; do
    if [ "$mm" = "0" ]; then export JXGPU_ROT_MISS_MAX=0; else unset JXGPU_ROT_MISS_MAX; fi
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --missing $rate 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('missing $rate rot_miss_max=$mm rotate_ms', round(s['rotate_k'],1), 'grm_ms', round(s['grm'],1), 'step_ms', round(d['ms_per_step'],1))"
  done
done
