"""Training path of the native DenoisingDiT: a `torch.autograd.Function` whose forward records a tape in device
memory (`bsi_dit_train_forward`) and whose backward is the hand-written HIP backward (`bsi_dit_backward`).
torch's autograd only carries the parameter gradients out; no torch op computes anything.

Replaces autograd over bsi/models/dit.py:87-103,174-181 of the reference inside
`BSI.train_loss(...).mean().backward()` (bsi/tasks/bsi.py:187-194)."""
import ctypes as C

import torch
from torch import Tensor

from .. import _native as N

_BLOCK_PARAMS = (("attn.to_qkv.weight", "qkv_w"), ("attn.to_qkv.bias", "qkv_b"), ("attn.to_out.weight", "out_w"),
                 ("attn.to_out.bias", "out_b"), ("mlp.0.weight", "fc1_w"), ("mlp.0.bias", "fc1_b"),
                 ("mlp.2.weight", "fc2_w"), ("mlp.2.bias", "fc2_b"), ("adaLN_modulation.0.weight", "ada0_w"),
                 ("adaLN_modulation.0.bias", "ada0_b"), ("adaLN_modulation.2.weight", "ada2_w"),
                 ("adaLN_modulation.2.bias", "ada2_b"))


def transposed_pack(model):
    """bf16 W^T shadows ([in][out]) of the block weights for the input-gradient GEMMs: persistent buffers that join the model's
    cast plan (`DenoisingDiT.native_pack`) on first use, after which ONE launch per parameter version refreshes both layouts."""
    key = model._weights_key()
    if getattr(model, "_pack_t", None) is not None and model._pack_t_key == key:
        return model._pack_t
    cfg, _, _, _ = model.native_pack()  # builds / refreshes the plan; if it already carries the transposes, they are current now
    if model._pack_t is not None and model._pack_t_key == key:
        return model._pack_t
    plan = model._plan
    dev = model.dit.patch_encoder.weight.device
    keep = []
    by_src = {d[0].data_ptr(): d for d in plan["descs"]}

    def shadow_t(w: Tensor):
        d = by_src[w.detach().data_ptr()]
        rows, cols = w.shape
        out = torch.empty((cols, rows), dtype=torch.bfloat16, device=dev)
        keep.append(out)
        d[2], d[6] = out, rows
        return out.data_ptr()

    blocks = (N.DitBlockWeightsT * cfg.depth)()
    for i, blk in enumerate(model.dit.blocks):
        b = blocks[i]
        b.qkv_wT = shadow_t(blk.attn.to_qkv.weight)
        b.out_wT = shadow_t(blk.attn.to_out.weight)
        b.fc1_wT = shadow_t(blk.mlp[0].weight)
        b.fc2_wT = shadow_t(blk.mlp[2].weight)
        b.ada2_wT = shadow_t(blk.adaLN_modulation[2].weight)
    wt = N.DitWeightsT()
    wt.blocks = C.cast(blocks, C.POINTER(N.DitBlockWeightsT))
    plan["pack_t"] = (wt, blocks, keep)
    model._finish_table(plan)
    table, n, tiles = plan["table"]
    with torch.no_grad():
        N.check(N.lib().bsi_cast_batch_bf16(N.ptr(table), n, tiles, N.stream()))  # first time: fills the transposes (and re-casts the rest)
    model._pack_t, model._pack_t_key = plan["pack_t"], key
    return model._pack_t


class _DitTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, drop_p, seed, mu, t, c_in, c_skip, c_out, *params):
        lib = N.lib()
        cfg, w, _, _ = model.native_pack()
        mu = mu.contiguous()
        t = t.detach().to(torch.float32).contiguous()
        B = mu.shape[0]
        out = torch.empty_like(mu)
        tape = torch.empty(lib.bsi_dit_tape_bytes(C.byref(cfg), B), dtype=torch.uint8, device=mu.device)
        N.check(lib.bsi_dit_train_forward(C.byref(cfg), C.byref(w), B, N.ptr(mu), N.ptr(t), N.ptr(c_in), N.ptr(c_skip),
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
        kpad = lib.bsi_dit_kpad(C.byref(cfg))
        named = dict(model.named_parameters())
        order = [n for n, _ in model.named_parameters()]
        # one flat fp32 buffer for all gradients; views are handed to autograd
        sizes = {n: named[n].numel() for n in order}
        total = sum(sizes.values())
        flat = getattr(model, "_grad_buffer", None)  # the data-parallel trainer's persistent (padded) gradient buffer, if any
        if flat is None:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
        assert flat.numel() >= total and flat.dtype == torch.float32 and flat.is_contiguous()
        views, off = {}, 0
        for n in order:
            views[n] = flat[off:off + sizes[n]].view_as(named[n])
            off += sizes[n]
        enc_pad = torch.empty((cfg.dim, kpad), dtype=torch.float32, device=dev)
        blocks = (N.DitBlockGrads * cfg.depth)()
        for i in range(cfg.depth):
            for pname, field in _BLOCK_PARAMS:
                setattr(blocks[i], field, views[f"dit.blocks.{i}.{pname}"].data_ptr())
        g = N.DitGrads()
        g.enc_w_padded = enc_pad.data_ptr()
        g.enc_b = views["dit.patch_encoder.bias"].data_ptr()
        g.dec_ln_w = views["dit.patch_decoder.0.weight"].data_ptr()
        g.dec_ln_b = views["dit.patch_decoder.0.bias"].data_ptr()
        g.dec_w = views["dit.patch_decoder.1.weight"].data_ptr()
        g.dec_b = views["dit.patch_decoder.1.bias"].data_ptr()
        g.blocks = C.cast(blocks, C.POINTER(N.DitBlockGrads))
        ws = torch.empty(lib.bsi_dit_backward_workspace_bytes(C.byref(cfg), B), dtype=torch.uint8, device=dev)
        # launch schedule of THIS backward (DPTrainer: CU reserve / tile queue while gradient buckets are in flight).  The switches are
        # per launching thread, and this function runs on autograd's device thread, not on the caller's: they are applied here, around
        # the one call whose launches they are meant for, and nothing else in the process sees them.
        reserve, queue = getattr(model, "_bwd_sched", None) or (0, False)
        try:
            if reserve:
                N.check(lib.bsi_set_cu_reserve(int(reserve)))
            if queue:
                N.check(lib.bsi_set_tile_queue(1))
            N.check(lib.bsi_dit_backward(C.byref(cfg), C.byref(w), C.byref(wt), C.byref(g), B, N.ptr(g_out), N.ptr(ctx.c_out),
                                         N.ptr(ctx.tape), N.ptr(ws), ctx.drop[0], ctx.drop[1], N.stream()))
        finally:
            if reserve:
                N.check(lib.bsi_set_cu_reserve(0))
            if queue:
                N.check(lib.bsi_set_tile_queue(0))
        ctx.tape = None
        model._last_flat_grad = flat  # the data-parallel trainer all-reduces and consumes this buffer directly
        kin = named["dit.patch_encoder.weight"].shape[1]
        views["dit.patch_encoder.weight"].copy_(enc_pad[:, :kin])
        if getattr(model, "_flat_grad_only", False):
            # the data-parallel trainer consumes `_last_flat_grad` itself: handing the views to autograd would make it copy every
            # one of them into a .grad tensor nobody reads (one copy kernel per parameter tensor)
            return (None,) * (8 + len(order))
        return (None, None, None, None, None, None, None, None, *[views[n] for n in order])


def dit_forward_train(model, mu: Tensor, t: Tensor, c_in=None, c_skip=None, c_out=None) -> Tensor:
    """x_hat = c_skip*mu + c_out*f(c_in*mu, t) (or f(mu, t)) with gradients w.r.t. the model parameters.
    In `train()` mode the blocks' dropout (attention weights and MLP input, dit.py:43-44,70,101) is applied with a
    counter-based mask whose seed derives from `torch.initial_seed()` and a per-model call counter (the reference
    draws these masks from the device's global generator, which cannot be reproduced bit for bit anyway)."""
    blk = model.dit.blocks[0]
    p = float(blk.attn.dropout) if model.training else 0.0
    seed = 0
    if p > 0.0:
        model._drop_calls = getattr(model, "_drop_calls", 0) + 1
        seed = (torch.initial_seed() * 0x9E3779B1 + model._drop_calls * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
    params = [q for _, q in model.named_parameters()]
    return _DitTrainFn.apply(model, p, seed, mu, t, c_in, c_skip, c_out, *params)


def dit_forward_autograd(model, mu: Tensor, t: Tensor) -> Tensor:
    return dit_forward_train(model, mu, t)
