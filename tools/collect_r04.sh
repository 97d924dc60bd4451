#!/usr/bin/env bash
# Round-4 evidence set, run ON THE GPU BOX from the repo root (gpurun):  tools/collect_r04.sh
#   gpurun_out/r04_bench_default.json / _full.json   the default `python bench.py` line (compact) and its full record
#   gpurun_out/r04_bench_all_extras.json             full record of `python bench.py --all-extras`
#   gpurun_out/prof_<tag>/                           rocprofv3 --kernel-trace --stats (all tags) + FETCH_SIZE / WRITE_SIZE passes (c2, c1, c2c, c5)
#   gpurun_out/r04_c2_sq_counters.txt                SQ wave-cycle breakdown on c2
# tools/profile_report.py <tag> r04 then turns the per-tag directories into profiles/r04_<tag>_kernel_stats.txt + profiles/pmc_traffic.json.
set -uo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
cp gpurun_out/bench_full.json gpurun_out/r04_bench_default_full.json
python bench.py --all-extras > gpurun_out/r04_bench_all_extras_line.json 2> gpurun_out/r04_bench_all_extras.err
cp gpurun_out/bench_full.json gpurun_out/r04_bench_all_extras.json
for tag in c2 c1 c2c c5; do
  tools/gpu_profile.sh $tag --workload $tag --no-extras --no-cpu-baseline $( [ $tag = c5 ] && echo "--steps 3 --warmup 1" ) > /dev/null 2>&1
done
for tag in c3 c4 zinc_json c4_mega c3_mega; do
  out="gpurun_out/prof_$tag"; mkdir -p "$out"
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o "$tag" -- python bench.py --workload $tag --no-extras --no-cpu-baseline > "$out/trace.log" 2>&1
done
sq="gpurun_out/pmc_sq_c2"; mkdir -p "$sq"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$sq/a" -o c2 -- python bench.py --no-extras --no-cpu-baseline > "$sq/a.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$sq/b" -o c2 -- python bench.py --no-extras --no-cpu-baseline > "$sq/b.log" 2>&1
{ echo "# rocprofv3 --pmc (two passes), fractions of SQ_WAVE_CYCLES per kernel, c2 (bench.py --no-extras): tools/pmc_sq_summary.py"; python tools/pmc_sq_summary.py "$sq/a"; echo; python tools/pmc_sq_summary.py "$sq/b"; } > gpurun_out/r04_c2_sq_counters.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete; find gpurun_out/pmc_sq_c2 -name "*counter_collection.csv" -delete
ls gpurun_out | head -60
