#!/usr/bin/env bash
# SQ instruction mix / issue cycles of the sweep kernels (four counters per pass): tools/pmc_sq_insts.sh <bench args...>
set -uo pipefail
export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
            "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); out="gpurun_out/pmc_insts/p$i"; mkdir -p "$out"
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o x -- python bench.py "$@" > "$out/log.txt" 2>&1
  python tools/pmc_raw_summary.py "$out" "agg_"
done
