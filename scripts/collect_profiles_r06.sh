#!/bin/bash
# Round-6 profile collection on the GPU box (run from the repo root through gpurun):
#   C3 (bench.py default shape): kernel-trace stats, HBM traffic counters (FETCH_SIZE and WRITE_SIZE in their own passes), MFMA
#   counters (clean panel and 1 % missing calls);
#   C4 (n = 50 000, m = 500 000 on one GPU): kernel stats + counter passes RESTRICTED to the Q2 / GRM / rotation kernels
#   (--kernel-include-regex: the unrestricted pass of round 4 did not finish inside 25 minutes);
#   C5 -BLUP PCG leg (bench.py --leg c5_pcg): kernel stats + FETCH_SIZE / WRITE_SIZE of the two streaming operator kernels.
# Counter passes never carry --kernel-trace / --stats (gpurun refuses the combination).  WHAT=c3|c4|c5|all (default all).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06
WHAT=${WHAT:-all}
mkdir -p $O
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE"
B3="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
if [ "$WHAT" = "all" ] || [ "$WHAT" = "c3" ]; then
  rm -rf $O/stats $O/*.json
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
  rm -f $O/stats/*/*kernel_trace.csv
  for pass in fetch:FETCH_SIZE: write:WRITE_SIZE: "fetch_fv:FETCH_SIZE:--mode fvlmm"; do
    name=${pass%%:*}; rest=${pass#*:}; ctr=${rest%%:*}; extra=${rest#*:}
    timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -- $B3 $extra > $O/$name.log 2>&1
    python3 scripts/pmc_summarize.py $O/$name $O/$name.json $name > /dev/null; rm -rf $O/$name
  done
  timeout 900 rocprofv3 --pmc $MF --output-format csv -d $O/mfma -- $B3 > $O/mfma.log 2>&1
  python3 scripts/pmc_summarize.py $O/mfma $O/mfma.json mfma > /dev/null; rm -rf $O/mfma
  timeout 900 rocprofv3 --pmc $MF --output-format csv -d $O/mfma_missing -- $B3 --missing 0.01 > $O/mfma_missing.log 2>&1
  python3 scripts/pmc_summarize.py $O/mfma_missing $O/mfma_missing.json mfma_missing > /dev/null; rm -rf $O/mfma_missing
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "c4" ]; then
  mkdir -p $O/c4; rm -rf $O/c4/*
  B4="python3 bench.py --samples 50000 --snps 500000 --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
  RX="sbback_apply|grm_i8_kernel|rotate_i8_"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4/stats -- $B4 > $O/c4_stats.log 2>&1
  rm -f $O/c4/stats/*/*kernel_trace.csv
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 700 rocprofv3 --pmc $ctr --kernel-include-regex "$RX" --output-format csv -d $O/c4/$name -- $B4 > $O/c4_$name.log 2>&1
    python3 scripts/pmc_summarize.py $O/c4/$name $O/c4/$name.json $name > /dev/null; rm -rf $O/c4/$name
  done
  timeout 700 rocprofv3 --pmc $MF --kernel-include-regex "$RX" --output-format csv -d $O/c4/mfma -- $B4 > $O/c4_mfma.log 2>&1
  python3 scripts/pmc_summarize.py $O/c4/mfma $O/c4/mfma.json mfma > /dev/null; rm -rf $O/c4/mfma
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "c5" ]; then
  mkdir -p $O/c5; rm -rf $O/c5/*
  B5="python3 bench.py --leg c5_pcg"
  RX5="pi_dot_kernel|pi_tdot_kernel"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5/stats -- $B5 > $O/c5_stats.log 2>&1
  rm -f $O/c5/stats/*/*kernel_trace.csv
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 600 rocprofv3 --pmc $ctr --kernel-include-regex "$RX5" --output-format csv -d $O/c5/$name -- $B5 > $O/c5_$name.log 2>&1
    python3 scripts/pmc_summarize.py $O/c5/$name $O/c5/$name.json $name > /dev/null; rm -rf $O/c5/$name
  done
fi
ls -la $O $O/stats/* $O/c4 $O/c5 2>/dev/null | head -60
