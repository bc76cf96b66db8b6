"""Golden fixture for the options of SelfAttentionBlock no shipped configuration turns on
(src/nn/attention.py:138-165, 255-290, 309-311), produced by the REFERENCE'S OWN module
imported verbatim on the hooks of make_golden.py, in float64 with float32-representable
inputs and parameters (forward, input gradients, parameter gradients):

  * "d": k_delta_rpe + q_delta_rpe next to the three edge-feature RPEs, in_proj / out_proj;
  * "s": k_delta_rpe shared with the queries (qk_share_rpe) on the NEGATED difference
         (q_on_minus_rpe), encoders shared by the heads, qk_scale='d+g';
  * "m": dropout on the attention weights with a FIXED keep-mask: the module's ``attn_drop``
         attribute is replaced by ``a -> a * mask / (1 - p)`` (the reference's forward is
         untouched and calls it where it calls nn.Dropout).

Usage (build container only): python tests/golden/make_golden_attention_options.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


class FixedDrop(torch.nn.Module):
    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, a):
        return a * self.mask.to(a.dtype) / (1 - self.p)


def run(tag, blk, gen, n, deg, c_in, rpe, out, mask_p=None):
    blk = blk.double()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(p.float().double())
    ei = mg.synth_graph(gen, n, deg)
    E = ei.shape[1]
    if mask_p is not None:
        mask = (torch.rand(E, blk.num_heads, generator=gen) >= mask_p)
        blk.attn_drop = FixedDrop(mask, mask_p)
        out[f"{tag}__mask"], out[f"{tag}__p"] = mask, mask_p
    x = mg.rnd(gen, n, c_in).requires_grad_()
    ea = mg.rnd(gen, E, rpe, scale=0.5).requires_grad_()
    y = blk(x, ei, edge_attr=ea)
    gw = mg.rnd(gen, *y.shape)
    (y * gw).sum().backward()
    out.update({f"{tag}__x": x, f"{tag}__edge_index": ei, f"{tag}__edge_attr": ea,
                f"{tag}__gw": gw, f"{tag}__out": y, f"{tag}__g_x": x.grad,
                f"{tag}__g_edge_attr": ea.grad})
    for k, p in blk.named_parameters():
        out[f"{tag}__p__{k}"] = p
        out[f"{tag}__g__{k}"] = p.grad


def main():
    U, N = mg.install_reference_import_hooks()
    gen = torch.Generator().manual_seed(777)
    torch.manual_seed(777)
    out = {}
    run("d", N.SelfAttentionBlock(64, num_heads=16, in_dim=48, out_dim=64, qk_dim=4,
                                  in_rpe_dim=18, k_rpe=True, q_rpe=True, v_rpe=True,
                                  k_delta_rpe=True, q_delta_rpe=True),
        gen, 83, 10.0, 48, 18, out)
    run("s", N.SelfAttentionBlock(32, num_heads=4, qk_dim=8, qk_scale="d+g", in_rpe_dim=7,
                                  k_rpe=True, q_rpe=True, k_delta_rpe=True, q_delta_rpe=True,
                                  qk_share_rpe=True, q_on_minus_rpe=True, heads_share_rpe=True),
        gen, 57, 8.0, 32, 7, out)
    run("m", N.SelfAttentionBlock(64, num_heads=16, out_dim=64, qk_dim=4, in_rpe_dim=32,
                                  k_rpe=True, q_rpe=True, v_rpe=True),
        gen, 71, 11.0, 64, 32, out, mask_p=0.3)
    mg.save("attention_options.npz", **out)


if __name__ == "__main__":
    main()
