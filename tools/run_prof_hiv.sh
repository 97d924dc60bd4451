# rocprofv3 kernel tables of the shipped HIV json layer at batch 128 (graph-block route + bit-mask dropout), captured step.
export TMPDIR=/tmp; mkdir -p gpurun_out
for tag in hiv_json_b128; do
  out="gpurun_out/prof_$tag"; mkdir -p "$out"
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o "$tag" -- python bench.py --workload $tag --no-extras --no-cpu-baseline --hipgraph --steps 200 --warmup 30 > "$out/trace.log" 2>&1
  tail -3 "$out/trace.log" | cut -c1-600
done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
find gpurun_out/prof_hiv_json_b128 -name "*kernel_stats.csv" | head
