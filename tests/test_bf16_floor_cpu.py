"""What the north_star tolerance (1e-2 relative on UNet eps) means for bf16: the fp32 oracle is re-run with ONLY the
matrix-multiply operands (activations and weights of every linear / conv, q, k, v and the softmax probabilities) rounded
to bf16 -- residual stream, normalisations, softmax and accumulation stay exact fp32.  That is the floor of ANY
implementation that feeds bf16 operands to the matrix cores, and it already sits above 1e-2 on the fixture the GPU
parity tests use (1.39e-2; the HIP path measures 1.8e-2 with its bf16 residual stream, fp16 operands 1.7e-3 / 2.5e-3).
So bf16 cannot meet 1e-2 here by keeping the residual stream in higher precision: the operand
rounding alone spends the budget.  fp16 (the reference's own dtype, `.half()`) is the mode held to 1e-2; bf16 is an
opt-in range-safe mode held to 2.5e-2 (tests/test_unet_gpu.py), i.e. < 2x this floor."""
from unittest import mock

import numpy as np
import os
import torch
import torch.nn.functional as F

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _emulated(dt, w, cfg, args):
    from oracle import nn_ops
    from oracle.unet3d import unet3d_forward
    r = lambda t: t.to(dt).float()
    lin0, conv0 = F.linear, F.conv2d

    def lin(x, wt, b=None):
        return lin0(r(x), r(wt), b)

    def conv(x, wt, b=None, **kw):
        return conv0(r(x), r(wt), b, **kw)

    def sdpa(q, k, v, heads):
        b, nq, c = q.shape
        d = c // heads
        qh, kh, vh = (r(t).reshape(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        s = (qh @ kh.transpose(-1, -2)) * d ** -0.5
        return (r(torch.softmax(s, -1)) @ vh).transpose(1, 2).reshape(b, nq, c)

    with mock.patch.object(F, "linear", lin), mock.patch.object(F, "conv2d", conv), mock.patch.object(nn_ops, "sdpa", sdpa):
        return unet3d_forward(w, cfg, *args)


def test_bf16_operand_rounding_alone_exceeds_1e2():
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights, unet3d_forward
    fx = np.load(os.path.join(G, "unet3d_v2_w64.npz"))
    cfg = UNet3DConfig.v2(block_out_channels=(64, 128, 256, 256))
    w = init_unet3d_weights(cfg, seed=int(fx["weight_seed"]))
    T = lambda a: torch.from_numpy(np.asarray(a))
    args = (T(fx["sample"]), int(fx["timestep"]), T(fx["ehs"]))
    ref = unet3d_forward(w, cfg, *args)
    assert ((ref - T(fx["out"])).norm() / ref.norm()).item() < 1e-5  # the oracle reproduces the reference fixture
    rel = {dt: ((_emulated(dt, w, cfg, args) - ref).norm() / ref.norm()).item() for dt in (torch.bfloat16, torch.float16)}
    assert 1.0e-2 < rel[torch.bfloat16] < 2.0e-2, rel   # measured 1.39e-2
    assert rel[torch.float16] < 3e-3, rel               # measured 1.7e-3
