timeout 900 python -m pytest tests/test_dc_hip.py tests/test_shipped_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
for u in 2048 0; do
  echo "dc_wave_units $u"
  for w in c4 c3; do
  DGN_DC_WAVE_UNITS=$u timeout 300 python bench.py --workload $w --steps 50 --warmup 10 --hipgraph --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$w captured ms',r['ms_per_step'])
"
  done
done
export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c4 -o c4 -- python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/prof_c4.log 2>&1
python tools/rocprof_top.py gpurun_out/prof_c4 16
bash tools/run_gb.sh c1
find gpurun_out/prof_c4 -name "*.db" -delete
