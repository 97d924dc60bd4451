#!/usr/bin/env bash
# Round-6 evidence set, run ON THE GPU BOX from the repo root (gpurun):  tools/collect_r06.sh
#   gpurun_out/r06_bench_default.json / _full.json   the default `python bench.py` line (compact) and its full record
#   gpurun_out/r06_bench_all_extras.json             full record of `python bench.py --all-extras`
#   gpurun_out/prof_<tag>/                           rocprofv3 --kernel-trace --stats (all tags) + FETCH_SIZE / WRITE_SIZE passes (c2, c3, c3_mega)
# tools/profile_report.py <tag> r06 then turns the per-tag directories into profiles/r06_<tag>_kernel_stats.txt + profiles/pmc_traffic.json.
set -uo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
cp gpurun_out/bench_full.json gpurun_out/r06_bench_default_full.json
for tag in c2 c3 c3_mega; do
  tools/gpu_profile.sh $tag --workload $tag --no-extras --no-cpu-baseline > /dev/null 2>&1
done
for tag in c1 c4 pattern_json zinc_json; do
  out="gpurun_out/prof_$tag"; mkdir -p "$out"
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o "$tag" -- python bench.py --workload $tag --no-extras --no-cpu-baseline > "$out/trace.log" 2>&1
done
out="gpurun_out/prof_c5_layer"; mkdir -p "$out"
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o c5_layer -- python bench.py --workload c5_layer --steps 3 --warmup 1 --no-extras --no-cpu-baseline > "$out/trace.log" 2>&1
for tag in c2_b128 zinc_json_b128 c1_b128 hiv_json_b128; do
  out="gpurun_out/prof_$tag"; mkdir -p "$out"
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o "$tag" -- python bench.py --workload $tag --no-extras --no-cpu-baseline --hipgraph --steps 200 --warmup 30 > "$out/trace.log" 2>&1
done
python tools/ab_option.py mix_bwd_fused 0 1 --kernels > gpurun_out/r06_c2_ab_mix_bwd_fused.txt 2>&1
python tools/ab_option.py bn_from_wgrad 0 1 > gpurun_out/r06_c2_ab_bn_from_wgrad.txt 2>&1
python tools/ab_option.py lin_wreg 0 3 --kernels > gpurun_out/r06_c2_ab_lin_wreg.txt 2>&1
python tools/ab_option.py bd_bwd_fused 0 1 --kernels > gpurun_out/r06_c2_ab_bd_bwd_fused.txt 2>&1
python tools/ab_option.py bn_stats_fused 0 1 --kernels > gpurun_out/r06_c2_ab_bn_stats_fused.txt 2>&1
python tools/hub_training_time.py > gpurun_out/r06_hub_training.txt 2>&1
python bench.py --all-extras > gpurun_out/r06_bench_all_extras_line.json 2> gpurun_out/r06_bench_all_extras.err
cp gpurun_out/bench_full.json gpurun_out/r06_bench_all_extras.json
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
du -sh gpurun_out; ls gpurun_out | head -60
