#!/usr/bin/env python3
"""Kernel table of the captured net step at the reference's batch size (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgn_amd import synth
from dgn_amd.nets import DGNNet
from dgn_amd.hipgraph import CapturedNetStep, bucket_capacity
dev = torch.device("cuda")
b = synth.molecule_batch(128, seed=41, extra_bonds=3.9, eig_dim=6)
N, E = int(b["num_nodes"]), b["src"].numel()
net = DGNNet(dict(num_atom_type=28, num_bond_type=4, hidden_dim=70, out_dim=70, in_feat_dropout=0.0, dropout=0.0, L=4, type_net="towers", pos_enc_dim=0,
                  readout="mean", graph_norm=True, batch_norm=True, aggregators="mean max min dir1-av dir1-dx", scalers="identity amplification attenuation",
                  avg_d={"log": torch.tensor(1.1)}, residual=True, edge_feat=False, edge_dim=0, pretrans_layers=1, posttrans_layers=1, device="cuda")).to(dev).train()
n_cap, e_cap = bucket_capacity(N, E)
cs = CapturedNetStep(net, n_cap, e_cap, 129, 6)
cs.load(b["src"].to(dev), b["dst"].to(dev), N, b["eig"].to(dev), torch.randint(0, 28, (N,)).to(dev), b["snorm_n"].to(dev), [int(s) for s in b["sizes"]], torch.randn(128, 1).to(dev))
cs.capture(3)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    cs.step()
torch.cuda.synchronize()
