import sys, time, torch
sys.path.insert(0, "/root/repo")
import dgn_amd
from dgn_amd import synth
for n in (128, 2048, 12000):
    b = synth.molecule_batch(n, seed=41, laplacian_eig=False)
    s, d, e = b["src"].cuda(), b["dst"].cuda(), b["eig"].cuda()
    N = int(b["num_nodes"])
    for _ in range(3):
        g = dgn_amd.DGNGraph(s, d, N, eig=e); g.ensure_csc()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        g = dgn_amd.DGNGraph(s, d, N, eig=e); g.ensure_csc()
    torch.cuda.synchronize()
    print(f"{n} graphs: N={N} E={s.numel()} DGNGraph + csc build {(time.perf_counter()-t0)/20*1e3:.3f} ms")
