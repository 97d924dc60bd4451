#!/usr/bin/env python3
"""Time the skinny fp32 GEMMs of the DGN layer with the BLAS back ends torch can reach."""
import os
import sys
import time
import torch

dev = torch.device("cuda:0")


def t_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    N = 275167
    cases = {
        "linear [N,70]x[70,140] (P|Q)": lambda: (torch.randn(N, 70, device=dev), torch.randn(140, 70, device=dev)),
        "linear [N,350]x[350,210] (block-diag posttrans)": lambda: (torch.randn(N, 350, device=dev), torch.randn(210, 350, device=dev)),
        "linear [N,150]x[150,225] (c1 posttrans)": lambda: (torch.randn(N, 150, device=dev), torch.randn(225, 150, device=dev)),
        "linear [N,350]x[350,70]": lambda: (torch.randn(N, 350, device=dev), torch.randn(70, 350, device=dev)),
    }
    if len(sys.argv) > 1 and sys.argv[1] == "tunable":
        import torch.cuda.tunable as tn
        tn.enable(True); tn.set_max_tuning_duration(200); tn.set_max_tuning_iterations(20)
        print("tunable op enabled")
    for lib in ("default",):
        if lib != "default":
            try:
                torch.backends.cuda.preferred_blas_library(lib)
            except Exception as e:
                print(lib, "unavailable", e); continue
        print("== preferred blas:", lib)
        for name, mk in cases.items():
            x, w = mk()
            ms = t_ms(lambda: torch.nn.functional.linear(x, w))
            fl = 2.0 * x.shape[0] * x.shape[1] * w.shape[0]
            by = 4.0 * (x.numel() + x.shape[0] * w.shape[0])
            print(f"  {name:50s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {by / ms / 1e6:7.0f} GB/s")
            # dW = dZ^T X  (reduction over N)
            dz = torch.randn(x.shape[0], w.shape[0], device=dev)
            ms = t_ms(lambda: dz.t() @ x)
            print(f"  {'   dW = dZ^T X':50s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
        # strided batched (towers): agg [N, 5, 70] -> [5, N, 70] view, w [5, 70, 42]
        agg = torch.randn(N, 5, 70, device=dev); w = torch.randn(5, 70, 42, device=dev)
        ms = t_ms(lambda: torch.bmm(agg.transpose(0, 1), w))
        print(f"  {'bmm towers [5,N,70]x[5,70,42] (strided)':50s} {ms:8.3f} ms")
        aggc = agg.transpose(0, 1).contiguous()
        ms = t_ms(lambda: torch.bmm(aggc, w))
        print(f"  {'bmm towers contiguous':50s} {ms:8.3f} ms")
        dz = torch.randn(5, N, 42, device=dev)
        ms = t_ms(lambda: torch.bmm(agg.transpose(0, 1).transpose(1, 2), dz))
        print(f"  {'bmm towers dW [5,70,N]x[5,N,42]':50s} {ms:8.3f} ms")
        ms = t_ms(lambda: torch.bmm(dz, w.transpose(1, 2)))
        print(f"  {'bmm towers dA [5,N,42]x[5,42,70]':50s} {ms:8.3f} ms")


if __name__ == "__main__":
    main()
