"""CPU oracle of the DiT denoiser (TEST INFRASTRUCTURE — see oracle/__init__.py).

Functional restatement of ``bsi/models/dit.py``, ``bsi/models/pos_emb.py``,
``bsi/nn/fourier_features.py`` and ``bsi/nn/mlp.py`` of the reference.  Takes a flat
dict of weights with the reference's state-dict keys.

``matmul_dtype`` emulates the HIP path's rounding points: with ``torch.bfloat16`` every
GEMM/attention operand is rounded to bf16 before the (fp32-accumulated) product, exactly
where the HIP kernels round — used to separate "kernel bug" from "bf16 rounding" in parity
tests.  With ``None`` it is the plain fp32/fp64 restatement.
"""
import math

import numpy as np
import torch


def nyquist_tables(size: int, expected_rate: int):
    """pos_emb.py:44-77 — (scale, bias) fp32 buffers."""
    assert size % 2 == 0
    k = size // 2
    nyquist = expected_rate / 2
    golden = (1 + np.sqrt(5)) / 2
    freqs = np.geomspace(1 / 8, nyquist / (2 * golden), num=k)
    scale = np.repeat(2 * np.pi * freqs, 2)
    bias = np.tile(np.array([0, np.pi / 2]), k)
    return torch.tensor(scale, dtype=torch.float32), torch.tensor(bias, dtype=torch.float32)


def nyquist_embedding(t, size, expected_rate):
    """pos_emb.py:78-84 — sin(bias + scale * t[..., None])."""
    scale, bias = nyquist_tables(size, expected_rate)
    scale, bias = scale.to(t.dtype), bias.to(t.dtype)
    return torch.addcmul(bias, scale, t[..., None]).sin()


def fourier_features(x, n_min, n_max, dim=1, table_dtype=torch.float32):
    """fourier_features.py:11-36 — channel order (input channel, n, offset).  The reference
    builds ``coefs``/``offsets`` in torch's DEFAULT dtype (fp32 in production, fp64 under its
    tests/conftest.py) — ``table_dtype`` restates that."""
    ns = torch.arange(n_min, n_max + 1)
    coefs = (2 * torch.pi * (2 ** ns).to(table_dtype)).to(x.dtype)
    offsets = torch.tensor([0, torch.pi / 2], dtype=table_dtype).to(x.dtype)
    right = x.dim() - dim - 1
    xx = x.unsqueeze(dim + 1).unsqueeze(dim + 1)
    args = torch.addcmul(offsets.view(-1, *([1] * right)), coefs.view(-1, *([1] * (right + 1))), xx)
    return args.sin().flatten(start_dim=dim, end_dim=dim + 2)


def patch_pos_embedding(hidden, H, W, ps):
    """dit.py:135-146."""
    ph, pw = H // ps, W // ps
    e_h = nyquist_embedding(torch.linspace(0, 1, ph), hidden // 2, max(H, W))
    e_w = nyquist_embedding(torch.linspace(0, 1, pw), hidden // 2, max(H, W))
    return torch.cat((e_h.repeat_interleave(pw, dim=0), e_w.repeat(ph, 1)), dim=1)


def patchify(x, ps):
    """dit.py:149-153: 'b c (nh ps_h) (nw ps_w) -> b (nh nw) (ps_h ps_w c)'."""
    B, C, H, W = x.shape
    nh, nw = H // ps, W // ps
    return x.reshape(B, C, nh, ps, nw, ps).permute(0, 2, 4, 3, 5, 1).reshape(B, nh * nw, ps * ps * C)


def unpatchify(x, ps, H, W):
    """dit.py:166-172: 'b (nh nw) (ps_h ps_w c) -> b c (nh ps_h) (nw ps_w)'."""
    B = x.shape[0]
    nh, nw = H // ps, W // ps
    C = x.shape[2] // (ps * ps)
    return x.reshape(B, nh, nw, ps, ps, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)


def layer_norm(x, eps=1e-5, weight=None, bias=None):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def gelu_tanh(x):
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def silu(x):
    return x * torch.sigmoid(x)


def _rt(x, md):
    return x if md is None else x.to(md).to(x.dtype)


def linear(x, w, b, md=None):
    return _rt(x, md) @ _rt(w, md).t() + b


def attention(x, w_qkv, b_qkv, w_out, b_out, heads, md=None, drop=None):
    """dit.py:36-47: qkv split '(qkv h c)', softmax(q k^T / sqrt(c)) v, merge '(h c)'."""
    B, N, d = x.shape
    c = d // heads
    qkv = linear(x, w_qkv, b_qkv, md).reshape(B, N, 3, heads, c).permute(2, 0, 3, 1, 4)
    q, k, v = _rt(qkv[0], md), _rt(qkv[1], md), _rt(qkv[2], md)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(c))
    p = torch.softmax(s, dim=-1)
    if drop is not None:  # F.scaled_dot_product_attention(dropout_p): keep-mask * 1/(1-p) on the weights
        p = p * drop
    o = _rt(p, md) @ v
    o = o.permute(0, 2, 1, 3).reshape(B, N, d)
    return linear(_rt(o, md), w_out, b_out, md)


def dit_block(x, c, W, pre, heads, md=None, drop_attn=None, drop_mlp=None):
    """dit.py:87-103."""
    h = linear(c, W[pre + "adaLN_modulation.0.weight"], W[pre + "adaLN_modulation.0.bias"], md)
    mod = linear(silu(h), W[pre + "adaLN_modulation.2.weight"], W[pre + "adaLN_modulation.2.bias"], md)
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    a_in = torch.addcmul(sh_a[:, None], sc_a[:, None] + 1, layer_norm(x))
    # with md set, branch outputs are rounded once more: the HIP engine hands them to the residual update as bf16
    x = torch.addcmul(x, g_a[:, None], _rt(attention(
        a_in, W[pre + "attn.to_qkv.weight"], W[pre + "attn.to_qkv.bias"],
        W[pre + "attn.to_out.weight"], W[pre + "attn.to_out.bias"], heads, md, drop_attn), md))
    m_in = torch.addcmul(sh_m[:, None], sc_m[:, None] + 1, layer_norm(x))
    if drop_mlp is not None:  # nn.Dropout in front of the MLP (dit.py:70,101)
        m_in = m_in * drop_mlp
    hdn = gelu_tanh(linear(m_in, W[pre + "mlp.0.weight"], W[pre + "mlp.0.bias"], md))
    x = torch.addcmul(x, g_m[:, None], _rt(linear(_rt(hdn, md), W[pre + "mlp.2.weight"], W[pre + "mlp.2.bias"], md), md))
    return x


def dit_forward(W, mu, t, *, patch_size, dim, depth, heads, ff=None, md=None, return_tokens=False, drop=None):
    """DenoisingDiT.forward (dit.py:225-233) + DiT.forward (dit.py:174-181).
    ``ff``: None or (n_min, n_max)."""
    B, C, H, Wd = mu.shape
    parts = [mu]
    if ff is not None:
        parts.append(fourier_features(mu, ff[0], ff[1], dim=1))
    x = torch.cat(parts, dim=1)
    c = nyquist_embedding(t, dim, 1000)
    pos = patch_pos_embedding(dim, H, Wd, patch_size).to(mu.dtype)
    x = linear(patchify(x, patch_size), W["dit.patch_encoder.weight"], W["dit.patch_encoder.bias"], md) + pos
    for i in range(depth):
        x = dit_block(x, c, W, f"dit.blocks.{i}.", heads, md,
                      None if drop is None else drop.get(("attn", i)), None if drop is None else drop.get(("mlp", i)))
    if return_tokens:
        return x
    y = layer_norm(x, 1e-5, W["dit.patch_decoder.0.weight"], W["dit.patch_decoder.0.bias"])
    y = linear(y, W["dit.patch_decoder.1.weight"], W["dit.patch_decoder.1.bias"], md)
    return unpatchify(y, patch_size, H, Wd)


def dit_param_shapes(data_shape, patch_size, dim, depth, ff=None):
    """State-dict contract."""
    C = data_shape[0]
    cin = C + (C * (ff[1] - ff[0] + 1) * 2 if ff is not None else 0)
    pa = patch_size ** 2
    shapes = {"dit.patch_encoder.weight": (dim, pa * cin), "dit.patch_encoder.bias": (dim,)}
    for i in range(depth):
        p = f"dit.blocks.{i}."
        shapes.update({
            p + "attn.to_qkv.weight": (3 * dim, dim), p + "attn.to_qkv.bias": (3 * dim,),
            p + "attn.to_out.weight": (dim, dim), p + "attn.to_out.bias": (dim,),
            p + "mlp.0.weight": (4 * dim, dim), p + "mlp.0.bias": (4 * dim,),
            p + "mlp.2.weight": (dim, 4 * dim), p + "mlp.2.bias": (dim,),
            p + "adaLN_modulation.0.weight": (dim, dim), p + "adaLN_modulation.0.bias": (dim,),
            p + "adaLN_modulation.2.weight": (6 * dim, dim), p + "adaLN_modulation.2.bias": (6 * dim,),
        })
    shapes.update({"dit.patch_decoder.0.weight": (dim,), "dit.patch_decoder.0.bias": (dim,),
                   "dit.patch_decoder.1.weight": (pa * C, dim), "dit.patch_decoder.1.bias": (pa * C,)})
    return shapes


def dit_random_weights(data_shape, patch_size, dim, depth, ff=None, seed=0, dtype=torch.float32,
                       adaln_std=0.02):
    """PyTorch-default-like init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))), with the adaLN output layer
    ~N(0, adaln_std^2) instead of zero so that blocks are not the identity."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shp in dit_param_shapes(data_shape, patch_size, dim, depth, ff).items():
        if name.startswith("dit.patch_decoder.0."):
            W[name] = torch.ones(shp, dtype=dtype) if name.endswith("weight") else torch.zeros(shp, dtype=dtype)
            if True:  # perturb so affine LN is exercised
                W[name] = W[name] + 0.05 * torch.randn(shp, generator=g, dtype=dtype)
        elif "adaLN_modulation.2" in name:
            W[name] = adaln_std * torch.randn(shp, generator=g, dtype=dtype)
        else:
            fan_in = shp[1] if len(shp) == 2 else None
            if fan_in is None:
                # bias: fan_in of the matching weight
                fan_in = W[name.replace("bias", "weight")].shape[1]
            bound = 1 / math.sqrt(fan_in)
            W[name] = (torch.rand(shp, generator=g, dtype=dtype) * 2 - 1) * bound
    return W
