#!/usr/bin/env python3
"""BatchNorm statistics / apply kernels on dense rows of a width that is no multiple of four: the flat 16-byte-chunk kernels (16-byte aligned
operands) against the thread-per-column(-pair) kernels (the same tensors 8 bytes off alignment take those).  us per call, HIP events."""
import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgn_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
st = _lib.stream_ptr(dev)

def timed(fn, n=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

for N, F in ((15000, 65), (15000, 47), (52754, 70), (275167, 75), (275167, 70)):
    res = {}
    for off in (0, 2):
        buf = lambda: torch.randn(N * F + 8, device=dev)[off:off + N * F].view(N, F)
        x, g, y, gx = buf(), buf(), buf(), buf()
        ga, be, rm, rv = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev), torch.zeros(F, device=dev), torch.ones(F, device=dev)
        mean, inv, gg, gb, sums = [torch.empty(F, device=dev) for _ in range(4)] + [torch.empty(2 * F, device=dev)]
        nb = lib.dgn_bn_tail_workspace_bytes(N, F)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())
        fwd_stats = lambda: lib.dgn_bn_tail_forward(N, F, P(x), F, P(ga), P(be), P(rm), P(rv), C.c_float(0.1), C.c_float(1e-5), 1, 1, None, None, P(mean), P(inv), P(ws), nb, None, st)
        fwd_all = lambda: lib.dgn_bn_tail_forward(N, F, P(x), F, P(ga), P(be), P(rm), P(rv), C.c_float(0.1), C.c_float(1e-5), 1, 1, P(g), P(y), P(mean), P(inv), P(ws), nb, None, st)
        bwd_stats = lambda: lib.dgn_bn_tail_backward(N, F, P(g), P(x), F, P(ga), P(be), P(mean), P(inv), 1, None, P(gg), P(gb), P(sums), P(ws), nb, None, st)
        assert fwd_all() == 0 and bwd_stats() == 0, lib.dgn_last_error()
        res[off] = (timed(fwd_stats), timed(fwd_all), timed(bwd_stats))
    print(f"[{N}, {F}]  stats+finalize {res[2][0]:.1f} -> {res[0][0]:.1f} us   stats+finalize+apply {res[2][1]:.1f} -> {res[0][1]:.1f}   bwd stats+finalize {res[2][2]:.1f} -> {res[0][2]:.1f}   (column kernels -> flat)")
