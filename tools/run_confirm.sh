# Final confirmation on a fresh box: GPU suite, smoke, the default bench line (run from the repo root through gpurun).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
cp gpurun_out/bench_full.json gpurun_out/r05_bench_default_full.json 2>/dev/null
cat gpurun_out/gpu_tests.txt gpurun_out/smoke.txt; head -c 3000 gpurun_out/r05_bench_default.json
