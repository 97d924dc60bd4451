#!/usr/bin/env bash
# SQ wave-cycle breakdown of a workload's kernels on the GPU box: tools/pmc_sq_tag.sh <tag>  ->  gpurun_out/<tag>_sq_counters.txt
set -uo pipefail
tag="$1"; export TMPDIR=/tmp
sq="gpurun_out/pmc_sq_$tag"; mkdir -p "$sq"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$sq/a" -o $tag -- python bench.py --workload $tag --no-extras --no-cpu-baseline > "$sq/a.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$sq/b" -o $tag -- python bench.py --workload $tag --no-extras --no-cpu-baseline > "$sq/b.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d "$sq/c" -o $tag -- python bench.py --workload $tag --no-extras --no-cpu-baseline > "$sq/c.log" 2>&1
{ echo "# rocprofv3 --pmc, fractions of SQ_WAVE_CYCLES per kernel, $tag (bench.py --no-extras): tools/pmc_sq_summary.py"; python tools/pmc_sq_summary.py "$sq/a"; echo; python tools/pmc_sq_summary.py "$sq/b"; echo; python tools/pmc_sq_summary.py "$sq/c"; } > gpurun_out/${tag}_sq_counters.txt 2>&1
find "$sq" -name "*counter_collection.csv" -delete; find "$sq" -name "*kernel_trace.csv" -delete; find "$sq" -name "*.db" -delete
