#!/usr/bin/env python3
"""dgn_gemm_* vs the library GEMM on the simple / complex layers' posttrans shapes: max error vs fp64, timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgn_amd import ops

dev = torch.device("cuda")
shapes = [("c1 posttrans", 275167, 152, 225), ("c4 posttrans", 52754, 350, 210), ("c3 posttrans", 14649, 198, 65), ("c2 complex", 275167, 420, 210)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps

for name, M, k, n in shapes:
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, k, device=dev, generator=gen)
    w = torch.randn(n, k, device=dev, generator=gen) / k ** 0.5
    b = torch.randn(n, device=dev, generator=gen)
    g = torch.randn(M, n, device=dev, generator=gen)
    res = {}
    for tag, f in (("lib", torch.nn.functional.linear), ("own", ops.wide_linear)):
        xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = f(xx, ww, bb)
        gx, gw, gb = torch.autograd.grad(y, [xx, ww, bb], g)
        res[tag] = (y.detach(), gx, gw, gb)
    sub = slice(0, 4096)
    y64 = x[sub].double() @ w.double().T + b.double()
    gx64 = g[sub].double() @ w.double()
    gw64 = g.double().T @ x.double()
    err = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print(name, "fwd err own/lib", err(res["own"][0][sub], y64), err(res["lib"][0][sub], y64), "dgrad", err(res["own"][1][sub], gx64), err(res["lib"][1][sub], gx64),
          "wgrad", err(res["own"][2], gw64), err(res["lib"][2], gw64))
    flops = 2.0 * M * k * n
    from dgn_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    c = torch.empty(M, n, device=dev); gx = torch.empty(M, k, device=dev); gw = torch.empty(n, k, device=dev)
    wt = w.t().contiguous()
    nb = lib.dgn_gemm_wgrad_workspace_bytes(M, k, n); ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    own = dict(fwd=lambda: lib.dgn_gemm_forward(M, k, n, x.data_ptr(), k, w.data_ptr(), k, 0, b.data_ptr(), c.data_ptr(), n, st),
               dgrad_kn=lambda: lib.dgn_gemm_forward(M, n, k, g.data_ptr(), n, w.data_ptr(), k, 1, None, gx.data_ptr(), k, st),
               dgrad_t=lambda: lib.dgn_gemm_forward(M, n, k, g.data_ptr(), n, wt.data_ptr(), n, 0, None, gx.data_ptr(), k, st),
               wgrad=lambda: lib.dgn_gemm_wgrad(M, k, n, g.data_ptr(), n, x.data_ptr(), k, gw.data_ptr(), k, None, ws.data_ptr(), nb, st))
    libf = dict(fwd=lambda: torch.nn.functional.linear(x, w, b), dgrad=lambda: g @ w, wgrad=lambda: g.t() @ x)
    with torch.no_grad():
        print("   lib: " + "  ".join(f"{k_} {t(f):.3f} ms ({flops / t(f) / 1e9:.0f} TF)" for k_, f in libf.items()))
        print("   own: " + "  ".join(f"{k_} {t(f):.3f} ms ({flops / t(f) / 1e9:.0f} TF)" for k_, f in own.items()))
