mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/gpu_tests.txt
bash tools/collect_r05.sh > gpurun_out/collect.log 2>&1
