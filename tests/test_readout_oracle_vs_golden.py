"""The readout / VirtualNode oracle must reproduce the fixtures generated from the reference (G8)."""
import numpy as np
import torch

from oracle import readout_oracle as ro


def test_readout_modes(golden):
    g = golden("g8_readouts")
    sizes, eig = g["sizes"].tolist(), torch.from_numpy(g["eig"])
    for mode in ("sum", "max", "mean", "directional", "directional_abs"):
        h = torch.from_numpy(g[f"readout/{mode}/h"]).requires_grad_(True)
        hg = ro.readout(h, sizes, mode, eig)
        np.testing.assert_allclose(hg.detach().numpy(), g[f"readout/{mode}/hg"], rtol=1e-6, atol=1e-6, err_msg=mode)
        gh, = torch.autograd.grad(hg, [h], torch.from_numpy(g[f"readout/{mode}/cot"]))
        np.testing.assert_allclose(gh.numpy(), g[f"readout/{mode}/gh"], rtol=1e-6, atol=1e-6, err_msg=mode)


def _close(a, ref, msg=""):
    """fp32 agreement relative to the array's scale (BatchNorm over 3 graphs produces gradients of mixed magnitudes)."""
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(a, ref, rtol=1e-4, atol=1e-5 * scale, err_msg=msg)


def test_virtual_node(golden):
    g = golden("g8_readouts")
    sizes = g["sizes"].tolist()
    for c in range(int(g["vn/n_cases"])):
        pre = f"vn/c{c}"
        vn_type, b_norm, residual = g[f"{pre}/cfg"].tolist()
        sd = {k.split("sd::")[1]: torch.from_numpy(g[k]).clone().requires_grad_(g[k].dtype == np.float32 and "running" not in k)
              for k in g.files if k.startswith(pre + "/sd::")}
        h = torch.from_numpy(g[f"{pre}/h"]).requires_grad_(True)
        vh = torch.from_numpy(g[f"{pre}/vn_h"]).requires_grad_(True)
        vn_out, h_out, stats = ro.virtual_node_forward(sd, h, vh, sizes, vn_type, bool(int(residual)), training=True)
        np.testing.assert_allclose(vn_out.detach().numpy(), g[f"{pre}/vn_out"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(h_out.detach().numpy(), g[f"{pre}/h_out"], rtol=1e-5, atol=1e-6)
        names = [k.split("gp::")[1] for k in g.files if k.startswith(pre + "/gp::")]
        grads = torch.autograd.grad([vn_out, h_out], [h, vh] + [sd["" + n] for n in names],
                                    [torch.from_numpy(g[f"{pre}/cot_v"]), torch.from_numpy(g[f"{pre}/cot_h"])])
        _close(grads[0].numpy(), g[f"{pre}/gh"])
        _close(grads[1].numpy(), g[f"{pre}/gvn"])
        for n, gr in zip(names, grads[2:]):
            _close(gr.numpy(), g[f"{pre}/gp::{n}"], n)
        if int(b_norm):
            np.testing.assert_allclose(stats[0].numpy(), g[f"{pre}/after::fc_layer.b_norm.running_mean"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(stats[1].numpy(), g[f"{pre}/after::fc_layer.b_norm.running_var"], rtol=1e-5, atol=1e-6)
