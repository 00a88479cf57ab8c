#!/bin/bash
# Round-end profile collection on the GPU box (run from the repo root through gpurun; bench.py default = BASELINE configs[2]):
#   kernel-trace stats, HBM traffic counters (FETCH_SIZE and WRITE_SIZE in their own passes), MFMA counters.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_end
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rm -f $O/stats/*/*kernel_trace.csv
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/fetch.log 2>&1
python3 scripts/pmc_summarize.py $O/fetch $O/fetch.json fetch > /dev/null; rm -rf $O/fetch
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/write.log 2>&1
python3 scripts/pmc_summarize.py $O/write $O/write.json write > /dev/null; rm -rf $O/write
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_fv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --mode fvlmm > $O/fetch_fv.log 2>&1
python3 scripts/pmc_summarize.py $O/fetch_fv $O/fetch_fv.json fetch_fv > /dev/null; rm -rf $O/fetch_fv
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/mfma.log 2>&1
python3 scripts/pmc_summarize.py $O/mfma $O/mfma.json mfma > /dev/null; rm -rf $O/mfma
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_missing -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --missing 0.01 > $O/mfma_missing.log 2>&1
python3 scripts/pmc_summarize.py $O/mfma_missing $O/mfma_missing.json mfma_missing > /dev/null; rm -rf $O/mfma_missing
# the sliced int8 GEMM and the per-SNP series kernel are part of the passes above (oz_mm_kernel, series_coef_kernel)
if [ "${WITH_C4:-0}" = "1" ]; then
  # BASELINE configs[3] on one GPU: kernel stats and matrix-pipe counters of one step (n = 50 000, m = 500 000)
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4/stats -- python3 bench.py --samples 50000 --snps 500000 --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/c4_stats.log 2>&1
  rm -f $O/c4/stats/*/*kernel_trace.csv
  timeout 1500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d $O/c4/mfma -- python3 bench.py --samples 50000 --snps 500000 --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/c4_mfma.log 2>&1
  python3 scripts/pmc_summarize.py $O/c4/mfma $O/c4/mfma.json mfma > /dev/null; rm -rf $O/c4/mfma
fi
ls -la $O $O/stats/*
