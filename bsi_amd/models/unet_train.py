"""Training path of the native DenoisingVDMUNet: a `torch.autograd.Function` whose forward records a tape in device
memory (`bsi_unet_train_forward`) and whose backward is the hand-written HIP backward (`bsi_unet_backward`).
torch's autograd only carries the parameter gradients out; no torch op computes anything.

Replaces autograd over bsi/models/vdm_unet.py:92-100, bsi/nn/simplified_unet.py:33-48, bsi/nn/residual_block.py:61-64 and
bsi/nn/attention.py:32-41 of the reference inside `BSI.train_loss(...).mean().backward()` (bsi/tasks/bsi.py:187-194)."""
import ctypes as C

import torch
from torch import Tensor, nn

from .. import _native as N


def _block_names(model):
    L = model._cfg_args["levels"]
    return ([f"u_net.downsampling_blocks.{i}.0." for i in range(L)] + ["u_net.center_block.0.", "u_net.center_block.2."] +
            [f"u_net.upsampling_blocks.{i}.0." for i in range(L)])


def _grad_plan(model, cfg, flat, key):
    """Where the backward engine writes: views of `flat` in named_parameters order, the ctypes gradient tables holding their
    addresses, the staging buffers of the stacked FiLM gradient and the padded pos_map.1 gradient, and the device table of the
    copies that scatter them (and the shared conv2 / skip bias gradients) to their parameters' places."""
    lib = N.lib()
    dev = flat.device
    dim, cd = cfg.dim, cfg.c_dim
    named = dict(model.named_parameters())
    order = list(named)
    total = sum(q.numel() for q in named.values())
    assert flat.numel() >= total and flat.dtype == torch.float32 and flat.is_contiguous()
    views, off = {}, 0
    for n in order:
        views[n] = flat[off:off + named[n].numel()].view_as(named[n])
        off += named[n].numel()
    names = _block_names(model)
    blocks_m = model._blocks()
    arr = (N.UNetResBlockGrads * len(names))()
    skip_bias = []
    for i, (pfx, rb) in enumerate(zip(names, blocks_m)):
        last = len(rb.layers) - 1
        has_skip = isinstance(rb.skip, nn.Conv2d)
        arr[i].gn_w, arr[i].gn_b = views[pfx + "layers.0.weight"].data_ptr(), views[pfx + "layers.0.bias"].data_ptr()
        # convolution weight gradients are written straight in the nn.Conv2d layout of the flat buffer
        arr[i].conv1_w, arr[i].conv1_b = views[pfx + "layers.2.weight"].data_ptr(), views[pfx + "layers.2.bias"].data_ptr()
        arr[i].conv2_w = views[pfx + f"layers.{last}.weight"].data_ptr()
        arr[i].conv2_b = views[pfx + f"layers.{last}.bias"].data_ptr()
        arr[i].skip_w = views[pfx + "skip.weight"].data_ptr() if has_skip else None
        if has_skip:
            skip_bias.append((pfx + "skip.bias", pfx + f"layers.{last}.bias"))
    F = len(names) * 2 * dim
    film_w = torch.empty((F, cd), dtype=torch.float32, device=dev)
    film_b = torch.empty(F, dtype=torch.float32, device=dev)
    pm1_pad = torch.empty((cd, 64), dtype=torch.float32, device=dev)
    g = N.UNetGrads()
    g.enc_w, g.enc_b = views["encode.weight"].data_ptr(), views["encode.bias"].data_ptr()
    g.dec_w, g.dec_b = views["decode.weight"].data_ptr(), views["decode.bias"].data_ptr()
    g.pm1_w_padded, g.pm1_b = pm1_pad.data_ptr(), views["pos_map.1.bias"].data_ptr()
    g.pm3_w, g.pm3_b = views["pos_map.3.weight"].data_ptr(), views["pos_map.3.bias"].data_ptr()
    g.film_w, g.film_b = film_w.data_ptr(), film_b.data_ptr()
    g.blocks = C.cast(arr, C.POINTER(N.UNetResBlockGrads))
    apfx = "u_net.center_block.1.fn."
    g.agn_w, g.agn_b = views[apfx + "0.weight"].data_ptr(), views[apfx + "0.bias"].data_ptr()
    g.aqkv_w, g.aqkv_b = views[apfx + "1.to_qkv.weight"].data_ptr(), views[apfx + "1.to_qkv.bias"].data_ptr()
    g.aout_w, g.aout_b = views[apfx + "1.to_out.weight"].data_ptr(), views[apfx + "1.to_out.bias"].data_ptr()
    dst = [views[sb] for sb, _ in skip_bias]  # out = skip(x) + layers(x): both biases receive the same gradient
    src = [views[cb] for _, cb in skip_bias]
    for i, pfx in enumerate(names):
        dst += [views[pfx + "project_onto_scale_shift.weight"], views[pfx + "project_onto_scale_shift.bias"]]
        src += [film_w[i * 2 * dim:(i + 1) * 2 * dim], film_b[i * 2 * dim:(i + 1) * 2 * dim]]
    plan = {"key": key, "flat": flat, "views": views, "order": order, "g": g, "arr": arr, "film_w": film_w, "film_b": film_b,
            "pm1_pad": pm1_pad, "copy_dst": dst, "copy_src": src, "copy_table": None}
    if key is not None:  # a persistent buffer: the copy jobs go to the device once (the upload is a synchronous host-to-device copy)
        descs, tiles = (N.CopyDesc * len(dst))(), 0
        for i, (d, s_) in enumerate(zip(dst, src)):
            assert d.is_contiguous() and s_.is_contiguous() and d.numel() == s_.numel()
            descs[i].src, descs[i].dst, descs[i].len, descs[i].tile0 = s_.data_ptr(), d.data_ptr(), d.numel(), tiles
            tiles += lib.bsi_copy_batch_tiles(d.numel())
        plan.update(copy_table=torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev), copy_n=len(dst), copy_tiles=tiles)
    return plan


def transposed_pack(model):
    """bf16 shadows for the input-gradient products (rotated conv weights, transposed Linear weights), cached per parameter
    version like `native_pack`: persistent buffers, one batched launch for all convolution weights."""
    key = model._weights_key()
    if getattr(model, "_pack_t", None) is not None and model._pack_t_key == key:
        return model._pack_t
    lib = N.lib()
    skey = model._storage_key()
    if getattr(model, "_plan_t", None) is None or model._plan_t["skey"] != skey:
        model._plan_t = _build_plan_t(model, skey)
    plan = model._plan_t
    with torch.no_grad():
        N.check(lib.bsi_conv_weight_pack_batch(N.ptr(plan["descs"]), plan["ndesc"], 1, N.stream()))
        torch.cat(plan["film_ws"], dim=0, out=plan["film_w_cat"])
        for src, rows, cols, dst in plan["lin_t"]:
            N.check(lib.bsi_cast_transpose_bf16(N.ptr(src), rows, cols, N.ptr(dst), rows, N.stream()))
    model._pack_t = plan["pack"]
    model._pack_t_key = key
    return model._pack_t


def _build_plan_t(model, skey):
    dev = model.encode.weight.device
    keep, descs, lin = [], [], []

    def conv_t(conv: nn.Conv2d):
        cout, cin, kh, kw = conv.weight.shape
        taps = kh * kw
        w = conv.weight.detach()
        assert w.is_contiguous()
        out = torch.empty((cin, taps * cout), dtype=torch.bfloat16, device=dev)
        descs.append(N.ConvPackDesc(w.data_ptr(), out.data_ptr(), cout, cin, taps, cin, taps * cout, 0))
        keep.extend([w, out])
        return out.data_ptr()

    def lin_t(w: Tensor):
        assert w.is_contiguous()
        rows, cols = w.shape
        out = torch.empty((cols, rows), dtype=torch.bfloat16, device=dev)
        lin.append((w, rows, cols, out))
        keep.extend([w, out])
        return out.data_ptr()

    blocks = model._blocks()
    arr = (N.UNetResBlockWeightsT * len(blocks))()
    for i, rb in enumerate(blocks):
        arr[i].conv1_wT = conv_t(rb.layers[2])
        arr[i].conv2_wT = conv_t(rb.layers[-1])
        arr[i].skip_wT = conv_t(rb.skip) if isinstance(rb.skip, nn.Conv2d) else None
    wt = N.UNetWeightsT()
    wt.blocks = C.cast(arr, C.POINTER(N.UNetResBlockWeightsT))
    att = model.u_net.center_block[1].fn[1]
    wt.aqkv_wT, wt.aout_wT = conv_t(att.to_qkv), conv_t(att.to_out)
    film_ws = [rb.project_onto_scale_shift.weight.detach() for rb in blocks]
    film_w_cat = torch.empty((sum(t.shape[0] for t in film_ws), film_ws[0].shape[1]), dtype=torch.float32, device=dev)
    wt.film_wT = lin_t(film_w_cat)
    wt.pm3_wT = lin_t(model.pos_map[3].weight.detach())
    darr = (N.ConvPackDesc * len(descs))(*descs)
    descs_dev = torch.frombuffer(bytearray(bytes(darr)), dtype=torch.uint8).to(dev)
    keep.extend([film_ws, film_w_cat, descs_dev])
    return {"skey": skey, "pack": (wt, arr, keep), "descs": descs_dev, "ndesc": len(descs), "lin_t": lin, "film_ws": film_ws,
            "film_w_cat": film_w_cat}


class _UNetTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, drop_p, seed, mu, t, c_in, c_skip, c_out, *params):
        lib = N.lib()
        cfg, w, _, _ = model.native_pack()
        mu = mu.contiguous()
        t = t.detach().to(torch.float32).contiguous()
        B = mu.shape[0]
        out = torch.empty_like(mu)
        tape = torch.empty(lib.bsi_unet_tape_bytes(C.byref(cfg), B), dtype=torch.uint8, device=mu.device)
        N.check(lib.bsi_unet_train_forward(C.byref(cfg), C.byref(w), B, N.ptr(mu), N.ptr(t), N.ptr(c_in), N.ptr(c_skip),
                                           N.ptr(c_out), N.ptr(out), N.ptr(tape), drop_p, seed, N.stream()))
        ctx.model, ctx.tape, ctx.c_out, ctx.B = model, tape, c_out, B
        ctx.drop = (drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = N.lib()
        model, B = ctx.model, ctx.B
        cfg, w, _, _ = model.native_pack()
        wt, _, _ = transposed_pack(model)
        dev = g_out.device
        g_out = g_out.contiguous()
        flat = getattr(model, "_grad_buffer", None)  # the data-parallel trainer's persistent (padded) gradient buffer, if any
        if flat is not None:
            # persistent buffer: the views, the gradient tables and the copy jobs are built once per (buffer, parameter storage) --
            # ~760 tensor views and ~170 descriptors per step otherwise, a few ms of host time in front of the backward's first launch
            key = (flat.data_ptr(), flat.numel(), model._storage_key())
            plan = getattr(model, "_plan_g", None)
            if plan is None or plan["key"] != key:
                plan = model._plan_g = _grad_plan(model, cfg, flat, key)
        else:
            total = sum(q.numel() for q in model.parameters())
            plan = _grad_plan(model, cfg, torch.empty(total, dtype=torch.float32, device=dev), None)
        flat, views, g = plan["flat"], plan["views"], plan["g"]
        ws = torch.empty(lib.bsi_unet_backward_workspace_bytes(C.byref(cfg), B), dtype=torch.uint8, device=dev)
        N.check(lib.bsi_unet_backward(C.byref(cfg), C.byref(w), C.byref(wt), C.byref(g), B, N.ptr(g_out), N.ptr(ctx.c_out),
                                      N.ptr(ctx.tape), N.ptr(ws), ctx.drop[0], ctx.drop[1], N.stream()))
        ctx.tape = None
        # the stacked FiLM gradients and the shared skip / conv2 bias gradients go to their parameters' places in the flat buffer:
        # one launch of bsi_copy_batch_f32 instead of 167 when the buffer is the trainer's (torch._foreach_copy_ issues a hipMemcpyAsync per pair on ROCm)
        if plan["copy_table"] is not None:
            N.check(lib.bsi_copy_batch_f32(N.ptr(plan["copy_table"]), plan["copy_n"], plan["copy_tiles"], N.stream()))
        else:  # a fresh gradient buffer per call (plain autograd): no table to upload, no synchronisation
            torch._foreach_copy_(plan["copy_dst"], plan["copy_src"])
        views["pos_map.1.weight"].copy_(plan["pm1_pad"][:, :views["pos_map.1.weight"].shape[1]])
        model._last_flat_grad = flat
        if getattr(model, "_flat_grad_only", False):
            # the data-parallel trainer consumes `_last_flat_grad` itself: handing the views to autograd would make it copy every
            # one of them into a .grad tensor nobody reads (one copy kernel per parameter tensor)
            return (None,) * (8 + len(plan["order"]))
        return (None, None, None, None, None, None, None, None, *[views[n] for n in plan["order"]])


def unet_forward_train(model, mu: Tensor, t: Tensor, c_in=None, c_skip=None, c_out=None) -> Tensor:
    """x_hat = c_skip*mu + c_out*f(c_in*mu, t) (or f(mu, t)) with gradients w.r.t. the model parameters.  In `train()` mode
    the residual blocks' nn.Dropout (residual_block.py:46) is applied with a counter-based mask seeded from
    `torch.initial_seed()` and a per-model call counter."""
    p = float(model._dropout or 0.0) if model.training else 0.0
    seed = 0
    if p > 0.0:
        model._drop_calls = getattr(model, "_drop_calls", 0) + 1
        seed = (torch.initial_seed() * 0x9E3779B1 + model._drop_calls * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
    params = [q for _, q in model.named_parameters()]
    return _UNetTrainFn.apply(model, p, seed, mu, t, c_in, c_skip, c_out, *params)
