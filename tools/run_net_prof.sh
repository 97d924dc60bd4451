export TMPDIR=/tmp
mkdir -p gpurun_out/prof_net
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_net/trace -o net -- python -c "
import argparse, sys, torch
sys.path.insert(0, '.')
import bench
r = bench.run_net(argparse.Namespace(net_capture=True), torch.device('cuda'), steps=100, warmup=2)
print(r['captured'])
" > gpurun_out/prof_net/trace.log 2>&1
python tools/prof_summary.py gpurun_out/prof_net/trace/net_kernel_stats.csv 70
find gpurun_out/prof_net -name "*kernel_trace.csv" -delete
