#!/usr/bin/env python3
"""dgn_amd.ops.linear (streaming MFMA kernels) against torch's library GEMMs on the headline layer's shapes:
correctness (fp64 anchor) and time of forward, input gradient and weight gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgn_amd import ops  # noqa: E402


def ev(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 275167
    dev = "cuda"
    shapes = [("pretrans P|Q (block-diagonal weights, 5 towers)", 1, 70, 140, "bd"), ("pretrans P|Q", 1, 70, 140, True), ("posttrans (towers)", 5, 84, 42, False), ("mixing", 1, 70, 70, False),
              ("simple h=64", 1, 128, 64, True)]
    for name, T, k, n, has_bias in shapes:
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(T, M, k, device=dev, generator=g)
        w = torch.randn(T, n, k, device=dev, generator=g) / k ** 0.5
        if has_bias == "bd":
            mask = torch.zeros(n, k, device=dev)
            for tw in range(5):
                mask[14 * tw:14 * tw + 14, 14 * tw:14 * tw + 14] = 1
                mask[70 + 14 * tw:70 + 14 * tw + 14, 14 * tw:14 * tw + 14] = 1
            w = w * mask
        b = torch.randn(T, n, device=dev, generator=g) if has_bias else None
        gy = torch.randn(T, M, n, device=dev, generator=g)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.linear(xr, wr, b)
        y.backward(gy)
        x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
        y64 = torch.bmm(x64, w64.transpose(1, 2)) + (b.double().unsqueeze(1) if b is not None else 0)
        y64.backward(gy.double())
        xt, wt = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yt = torch.bmm(xt, wt.transpose(1, 2)) + (b.unsqueeze(1) if b is not None else 0)
        yt.backward(gy)
        err = lambda a_, r_: float((a_.double() - r_).abs().max() / r_.abs().max())
        print(f"{name}: T={T} M={M} k={k} n={n}")
        print(f"   rel err vs fp64   ours: y {err(y, y64):.2e} gx {err(xr.grad, x64.grad):.2e} gw {err(wr.grad, w64.grad):.2e}"
              f"   torch: y {err(yt, y64):.2e} gx {err(xt.grad, x64.grad):.2e} gw {err(wt.grad, w64.grad):.2e}")
        lib_fwd = lambda: torch.bmm(x, w.transpose(1, 2))
        lib_dg = lambda: torch.bmm(gy, w)
        lib_wg = lambda: torch.bmm(gy.transpose(1, 2), x)
        from dgn_amd import _lib
        L = _lib.load()
        ours_fwd = lambda: ops._lin_fwd(L, x, w, False, b, n)
        ours_dg = lambda: ops._lin_fwd(L, gy, w, True, None, k)
        ws_bytes = L.dgn_linear_wgrad_workspace_bytes(M, k, n, T)
        ws = torch.empty(max(ws_bytes // 4, 1), device=dev)
        gw = torch.empty_like(w)
        st = torch.cuda.current_stream().cuda_stream

        def ours_wg():
            _lib.check(L.dgn_linear_wgrad(M, k, n, T, gy.data_ptr(), n, gy.stride(0), x.data_ptr(), k, x.stride(0), gw.data_ptr(), k,
                                          n * k, None, 0, ws.data_ptr(), ws_bytes, st), "wgrad")
        byts = 4 * T * M * (k + n)
        for tag, ours, lib in (("forward", ours_fwd, lib_fwd), ("dgrad", ours_dg, lib_dg), ("wgrad", ours_wg, lib_wg)):
            to, tl = ev(ours), ev(lib)
            print(f"   {tag:8s} ours {to:7.1f} us ({byts / to / 1e6:5.2f} TB/s, {2 * T * M * k * n / to / 1e6:5.1f} TFLOP/s)   library {tl:7.1f} us")


if __name__ == "__main__":
    main()
