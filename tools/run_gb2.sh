timeout 900 python -m pytest tests/test_block_backward_gpu.py -x -q -m gpu 2>&1 | tail -3
for t in 0 1 2; do echo "persistent $t"; DGN_BLK_PERSISTENT=$t bash tools/run_gb.sh c2 c1 c2c zinc_json c4_mega; done
