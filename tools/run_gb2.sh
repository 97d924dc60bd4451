timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
bash tools/run_gb.sh c3 c3_mega pattern_json
cp gpurun_out/bench_full.json gpurun_out/pattern_json_full.json
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/gpu_pmc.sh c3mega_$c $c --workload c3_mega --steps 3 --warmup 1 --no-cpu-baseline --no-extras | grep -E "agg_|seg_sum"
done
