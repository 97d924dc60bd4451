# item 8 evidence: SQ instruction mix, wave-cycle breakdown and memory-pipeline counters of the c2 sweeps
export TMPDIR=/tmp
mkdir -p gpurun_out
args="--workload c2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
bash tools/pmc_sq_insts.sh $args > gpurun_out/r05_c2_sq_insts.txt 2>&1
sq="gpurun_out/pmc_sq_c2"; mkdir -p "$sq"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$sq/a" -o c2 -- python bench.py $args > "$sq/a.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAIT_INST_LDS --output-format csv -d "$sq/b" -o c2 -- python bench.py $args > "$sq/b.log" 2>&1
{ python tools/pmc_sq_summary.py "$sq/a"; echo; python tools/pmc_sq_summary.py "$sq/b"; } > gpurun_out/r05_c2_sq_counters.txt 2>&1
bash tools/pmc_multi.sh "agg_bwd_block|agg_fwd_short" $args > gpurun_out/r05_c2_mem_pipeline.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*counter_collection.csv" -delete
