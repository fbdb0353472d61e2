"""CPU oracle of the VDM-UNet denoiser (TEST INFRASTRUCTURE — see oracle/__init__.py).

Functional restatement of ``bsi/models/vdm_unet.py``, ``bsi/nn/residual_block.py``,
``bsi/nn/attention.py`` and ``bsi/nn/simplified_unet.py`` of the reference, driven by a flat
dict of weights with the reference's state-dict keys.
"""
import math

import torch
import torch.nn.functional as F

from .dit_oracle import _rt, fourier_features, linear, nyquist_embedding, silu


def group_norm(x, groups, weight, bias, eps=1e-5):
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(B, C, H, W)
    return y * weight[None, :, None, None] + bias[None, :, None, None]


def conv(x, w, b, md=None):
    """Conv2d, stride 1, zero padding k//2 (vdm_unet.py:71-72, residual_block.py:40-48)."""
    return F.conv2d(_rt(x, md), _rt(w, md), b, padding=w.shape[-1] // 2)


def residual_block(x, c, W, pre, *, has_dropout_slot, md=None, drop=None):
    """residual_block.py:61-64 with layers of :41-49.  The second conv sits at index 6 when a
    Dropout module occupies index 5 (dropout is not None), else at index 5.  ``drop``: optional dict
    {block prefix: keep-mask / (1 - p), [B, C, H, W]} applied where nn.Dropout sits (:46)."""
    i2 = 6 if has_dropout_slot else 5
    ss = linear(c, W[pre + "project_onto_scale_shift.weight"], W[pre + "project_onto_scale_shift.bias"], md)
    scale, shift = ss.chunk(2, dim=1)
    h = silu(group_norm(x, 32, W[pre + "layers.0.weight"], W[pre + "layers.0.bias"]))
    h = conv(h, W[pre + "layers.2.weight"], W[pre + "layers.2.bias"], md)
    h = torch.addcmul(shift[..., None, None], scale[..., None, None] + 1, h)  # FeatureModulation :21-24
    h = silu(h)
    if drop is not None:
        h = h * drop[pre]
    h = conv(h, W[pre + f"layers.{i2}.weight"], W[pre + f"layers.{i2}.bias"], md)
    skip = x
    if (pre + "skip.weight") in W:
        skip = conv(x, W[pre + "skip.weight"], W[pre + "skip.bias"], md)
    return skip + h


def attention2d(x, W, pre, heads, md=None):
    """attention.py:32-41: qkv channels '(qkv h c)', positions '(x y)'."""
    B, C, H, Wd = x.shape
    qkv = conv(x, W[pre + "to_qkv.weight"], W[pre + "to_qkv.bias"], md)
    c = C // heads
    qkv = qkv.reshape(B, 3, heads, c, H * Wd).permute(1, 0, 2, 4, 3)
    q, k, v = _rt(qkv[0], md), _rt(qkv[1], md), _rt(qkv[2], md)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(c))
    p = torch.softmax(s, dim=-1)
    o = _rt(p, md) @ v  # [B, h, P, c]
    o = o.permute(0, 1, 3, 2).reshape(B, C, H, Wd)
    return conv(o, W[pre + "to_out.weight"], W[pre + "to_out.bias"], md)


def unet_forward(W, mu, t, *, levels, pos_emb_size=32, pos_emb_rate=100, heads=1, ff=None,
                 has_dropout_slot=True, md=None, drop=None):
    """DenoisingVDMUNet.forward (vdm_unet.py:92-100) + SimplifiedUNet.forward (simplified_unet.py:33-48)."""
    parts = [mu]
    if ff is not None:
        parts.append(fourier_features(mu, ff[0], ff[1], dim=1))
    x = torch.cat(parts, dim=1)
    e = nyquist_embedding(t, pos_emb_size, pos_emb_rate)
    c = silu(linear(e, W["pos_map.1.weight"], W["pos_map.1.bias"], md))
    c = silu(linear(c, W["pos_map.3.weight"], W["pos_map.3.bias"], md))
    kw = dict(has_dropout_slot=has_dropout_slot, md=md, drop=drop)
    x = conv(x, W["encode.weight"], W["encode.bias"], md)
    skips = []
    for i in range(levels):
        x = residual_block(x, c, W, f"u_net.downsampling_blocks.{i}.0.", **kw)
        skips.append(x)
    x = residual_block(x, c, W, "u_net.center_block.0.", **kw)
    a = group_norm(x, 32, W["u_net.center_block.1.fn.0.weight"], W["u_net.center_block.1.fn.0.bias"])
    x = x + attention2d(a, W, "u_net.center_block.1.fn.1.", heads, md)
    x = residual_block(x, c, W, "u_net.center_block.2.", **kw)
    for i in range(levels):
        x = residual_block(torch.cat((x, skips.pop()), dim=1), c, W, f"u_net.upsampling_blocks.{i}.0.", **kw)
    return conv(x, W["decode.weight"], W["decode.bias"], md)


def unet_param_shapes(data_shape, dim, levels, pos_emb_size=32, pos_emb_mult=4, ff=None,
                      has_dropout_slot=True):
    C = data_shape[0]
    cin = C + (C * (ff[1] - ff[0] + 1) * 2 if ff is not None else 0)
    c_dim = pos_emb_size * pos_emb_mult
    i2 = 6 if has_dropout_slot else 5
    S = {"pos_map.1.weight": (c_dim, pos_emb_size), "pos_map.1.bias": (c_dim,),
         "pos_map.3.weight": (c_dim, c_dim), "pos_map.3.bias": (c_dim,),
         "encode.weight": (dim, cin, 3, 3), "encode.bias": (dim,),
         "decode.weight": (C, dim, 1, 1), "decode.bias": (C,)}

    def res(pre, din):
        S[pre + "project_onto_scale_shift.weight"] = (2 * dim, c_dim)
        S[pre + "project_onto_scale_shift.bias"] = (2 * dim,)
        S[pre + "layers.0.weight"] = (din,)
        S[pre + "layers.0.bias"] = (din,)
        S[pre + "layers.2.weight"] = (dim, din, 3, 3)
        S[pre + "layers.2.bias"] = (dim,)
        S[pre + f"layers.{i2}.weight"] = (dim, dim, 3, 3)
        S[pre + f"layers.{i2}.bias"] = (dim,)
        if din != dim:
            S[pre + "skip.weight"] = (dim, din, 1, 1)
            S[pre + "skip.bias"] = (dim,)

    for i in range(levels):
        res(f"u_net.downsampling_blocks.{i}.0.", dim)
    res("u_net.center_block.0.", dim)
    S["u_net.center_block.1.fn.0.weight"] = (dim,)
    S["u_net.center_block.1.fn.0.bias"] = (dim,)
    S["u_net.center_block.1.fn.1.to_qkv.weight"] = (3 * dim, dim, 3, 3)
    S["u_net.center_block.1.fn.1.to_qkv.bias"] = (3 * dim,)
    S["u_net.center_block.1.fn.1.to_out.weight"] = (dim, dim, 3, 3)
    S["u_net.center_block.1.fn.1.to_out.bias"] = (dim,)
    res("u_net.center_block.2.", dim)
    for i in range(levels):
        res(f"u_net.upsampling_blocks.{i}.0.", 2 * dim)
    return S


def unet_random_weights(data_shape, dim, levels, seed=0, dtype=torch.float32, **kw):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shp in unet_param_shapes(data_shape, dim, levels, **kw).items():
        if len(shp) == 1 and name.replace("bias", "weight") in W and len(W[name.replace("bias", "weight")].shape) == 1 \
                or (len(shp) == 1 and name.endswith("weight")):
            # GroupNorm affine
            base = torch.ones(shp, dtype=dtype) if name.endswith("weight") else torch.zeros(shp, dtype=dtype)
            W[name] = base + 0.05 * torch.randn(shp, generator=g, dtype=dtype)
        else:
            if name.endswith("weight"):
                fan_in = math.prod(shp[1:])
            else:
                fan_in = math.prod(W[name.replace("bias", "weight")].shape[1:])
            bound = 1 / math.sqrt(fan_in)
            W[name] = (torch.rand(shp, generator=g, dtype=dtype) * 2 - 1) * bound
    return W
