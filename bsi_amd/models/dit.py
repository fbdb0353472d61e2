"""Mirror of bsi/models/dit.py:26-233 (Attention, DiTBlock, DiT, DenoisingDiT) on the native DiT engine.

The module tree, constructor signatures, parameter names/shapes and non-persistent buffers are those
of the reference, so `state_dict`s interchange.  The torch submodules only HOLD the
fp32 parameters; `forward` runs `bsi_dit_adaln` + `bsi_dit_forward` (include/bsi_hip.h): bf16 MFMA GEMMs
with fused epilogues, fused attention, fp32 residual stream.  No torch compute op is on the path.
"""
import ctypes as C
from functools import partial

import torch
from torch import Tensor, nn

from .. import _native as N
from ..nn import FourierFeatures
from .pos_emb import NyquistPositionalEmbedding


# ---- CU-partitioned stream pair (bsi_dit_forward_pair, include/bsi_hip.h) ---------------------------------------------------------
# One pair per (device, h_cus), created on first use and kept for the life of the process (two HIP streams + an event ring).
_PAIRS: dict = {}
PAIR_MIN_BATCH = 64  # below this the GEMMs no longer fill the large partition with whole tiles: one chain on the caller's stream


def _pair_default():
    """(h_cus, flags) from BSI_CU_PAIR="<h_cus>[,attn_h]", or None (one stream).  Read per call: experiments flip it."""
    import os
    e = os.environ.get("BSI_CU_PAIR", "")
    if not e or e == "0":
        return None
    parts = e.split(",")
    return int(parts[0]), (1 if len(parts) > 1 and parts[1] == "attn_h" else 0)


def cu_pair_handle(device, h_cus: int):
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, h_cus)
    if key not in _PAIRS:
        h = C.c_void_p()
        with torch.cuda.device(idx):
            N.check(N.lib().bsi_cu_pair_create(h_cus, C.byref(h)))
        _PAIRS[key] = h
    return _PAIRS[key]


class Attention(nn.Module):
    """Parameter holder for dit.py:26-47 (`to_qkv`, `to_out`); rows of to_qkv are ordered (qkv, head, channel)."""

    def __init__(self, dim: int, *, heads: int, dropout: float = 0.0):
        super().__init__()
        self.heads = heads
        self.dropout = dropout
        self.to_qkv = nn.Linear(dim, dim * 3)
        self.to_out = nn.Linear(dim, dim)


class DiTBlock(nn.Module):
    """Parameter holder for dit.py:58-103 (adaLN-Zero block with the extra Linear before SiLU)."""

    def __init__(self, size: int, heads: int, mlp_ratio: int = 4, dropout: float | None = None):
        super().__init__()
        self.norm = nn.LayerNorm(size, elementwise_affine=False)
        self.attn = Attention(size, heads=heads, dropout=dropout if dropout is not None else 0.0)
        self.dropout = nn.Dropout(dropout) if dropout is not None else nn.Identity()
        # bsi.nn.MLP(in, out, hidden_features=[4*size], actfn=GELU(tanh)) is nn.Sequential(Linear, GELU, Linear)
        self.mlp = nn.Sequential(nn.Linear(size, mlp_ratio * size), nn.GELU(approximate="tanh"),
                                 nn.Linear(mlp_ratio * size, size))
        self.adaLN_modulation = nn.Sequential(nn.Linear(size, size), nn.SiLU(), nn.Linear(size, 6 * size))
        nn.init.constant_(self.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.adaLN_modulation[-1].bias, 0)


class DiT(nn.Module):
    """Parameter/buffer holder for dit.py:106-181."""

    def __init__(self, input_size, patch_size: int, in_channels: int, out_channels: int, hidden_size: int,
                 depth: int, heads: int, mlp_ratio: int, dropout: float | None):
        super().__init__()
        self.input_size = tuple(input_size)
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.out_channels = out_channels
        height, width = input_size
        patch_area = patch_size**2
        patches_h = height // patch_size
        patches_w = width // patch_size

        # fixed Fourier positional embeddings (dit.py:135-146), built on the host at construction
        pe = NyquistPositionalEmbedding(hidden_size // 2, max(height, width))

        def table(pos):
            return torch.addcmul(pe.bias, pe.scale, pos[..., None]).sin()

        pos_h = table(torch.linspace(0, 1, patches_h))
        pos_w = table(torch.linspace(0, 1, patches_w))
        pos_embs = torch.cat((pos_h.repeat_interleave(patches_w, dim=0), pos_w.repeat(patches_h, 1)), dim=1)
        self.register_buffer("patch_pos_embedding", pos_embs, persistent=False)
        self.t_embedding = NyquistPositionalEmbedding(hidden_size, 1000)

        self.patch_encoder = nn.Linear(patch_area * in_channels, hidden_size)
        self.blocks = nn.ModuleList(
            [DiTBlock(hidden_size, heads, mlp_ratio=mlp_ratio, dropout=dropout) for _ in range(depth)])
        self.patch_decoder = nn.Sequential(nn.LayerNorm(hidden_size), nn.Linear(hidden_size, patch_area * out_channels))


class DenoisingDiT(nn.Module):
    """Diffusion Transformer denoiser f(mu, t) — same constructor as bsi.models.dit.DenoisingDiT."""

    def __init__(self, data_shape, patch_size: int, dim: int, depth: int, heads: int, dropout: float | None = None,
                 fourier_features: FourierFeatures | None = None, **kwargs):
        super().__init__()
        self.data_shape = tuple(data_shape)
        self.fourier_features = fourier_features
        assert len(self.data_shape) == 3, "Only works for 2D images"
        n_channels = data_shape[0]
        in_channels = out_channels = n_channels
        if fourier_features is not None:
            in_channels += n_channels * fourier_features.n_features()
        self.dit = DiT(input_size=data_shape[1:], patch_size=patch_size, in_channels=in_channels,
                       out_channels=out_channels, hidden_size=dim, depth=depth, heads=heads, mlp_ratio=4,
                       dropout=dropout)
        self._cfg_args = dict(patch=patch_size, dim=dim, depth=depth, heads=heads)
        self._pack = None       # cached bf16 weight shadows + ctypes tables
        self._pack_key = None
        self._ws = None         # cached workspace tensor
        self._pack_t = None     # cached transposed shadows (training)
        self._pack_t_key = None
        self._plan = None       # persistent shadow buffers + descriptor table of the one-launch cast
        self.cu_pair = "env"    # (h_cus, flags) | None | "env" (= BSI_CU_PAIR): two CU-masked streams for large inference batches

    # ------------------------------------------------------------------------------------------------
    # native plumbing
    # ------------------------------------------------------------------------------------------------
    def _config(self) -> N.DitConfig:
        Cc, H, W = self.data_shape
        ff = self.fourier_features
        a = self._cfg_args
        return N.DitConfig(Cc, H, W, a["patch"], a["dim"], a["depth"], a["heads"],
                           ff.n_min if ff is not None else 1, ff.n_max if ff is not None else 0)

    _NATIVE_CACHES = ("_pack", "_pack_key", "_pack_t", "_pack_t_key", "_plan", "_plan_t", "_ws", "_last_flat_grad", "_grad_buffer",
                      "_params_pending_sync", "_bwd_sched")

    def __deepcopy__(self, memo):
        """`copy.deepcopy(model)` (EMA copies, checkpoint tooling) after the model has run: the native caches hold ctypes tables with raw
        device pointers into THIS model's buffers -- they are neither picklable nor valid for a copy, and are rebuilt on first use."""
        import copy
        saved = {k: self.__dict__.pop(k) for k in self._NATIVE_CACHES if k in self.__dict__}
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
            for k in saved:
                new.__dict__[k] = None
        finally:
            self.__dict__.update(saved)
        return new

    def _weights_key(self):
        ps = list(self.parameters())
        return (ps[0].device, ps[0].data_ptr(), sum(p._version for p in ps))

    def _storage_key(self):
        """Identity of the parameters' STORAGE: the pack plan bakes every parameter's raw pointer into a device-side descriptor table
        and into the ctypes weight tables, so replacing the storage of a parameter must rebuild the plan."""
        ps = list(self.parameters())
        return (ps[0].device, len(ps), hash(tuple(p.data_ptr() for p in ps)))

    def _build_plan(self, skey):
        """Persistent bf16 shadows ([N][K] row-major) of the GEMM weights, the ctypes weight table pointing at them and at the fp32
        parameters, and the descriptor list of the one-launch cast (`bsi_cast_batch_bf16`).  No device work here."""
        lib = N.lib()
        dev = self.dit.patch_encoder.weight.device
        if dev.type != "cuda":
            raise RuntimeError("bsi_amd.DenoisingDiT: parameters must live on a HIP device (no CPU path)")
        cfg = self._config()
        kpad = lib.bsi_dit_kpad(C.byref(cfg))
        keep, descs = [], []

        def dense(q: Tensor) -> Tensor:
            """The descriptor table and the weight tables hold raw pointers and assume row stride = cols: a strided parameter
            (a transposed view, a loaded slice) is made contiguous IN PLACE once, before its pointer is taken."""
            if not q.is_contiguous():
                q.data = q.data.contiguous()
            return q.detach()

        def shadow(w: Tensor, ld=None, out=None):
            w = dense(w)
            rows, cols = w.shape
            ld = ld or cols
            if out is None:
                out = torch.empty((rows, ld), dtype=torch.bfloat16, device=dev)
            keep.extend([w, out])
            descs.append([w, out, None, rows, cols, ld, 0])
            return out.data_ptr()

        def f32(p: Tensor):
            t = dense(p)
            keep.append(t)
            return t.data_ptr()

        # the adaLN matrices of all blocks in one buffer each: uniform strides let the engine run them as one grouped GEMM
        dim_ = self._cfg_args["dim"]
        ada0_all = torch.empty((cfg.depth, dim_, dim_), dtype=torch.bfloat16, device=dev)
        ada2_all = torch.empty((cfg.depth, 6 * dim_, dim_), dtype=torch.bfloat16, device=dev)
        blocks = (N.DitBlockWeights * cfg.depth)()
        for i, blk in enumerate(self.dit.blocks):
            b = blocks[i]
            b.qkv_w, b.qkv_b = shadow(blk.attn.to_qkv.weight), f32(blk.attn.to_qkv.bias)
            b.out_w, b.out_b = shadow(blk.attn.to_out.weight), f32(blk.attn.to_out.bias)
            b.fc1_w, b.fc1_b = shadow(blk.mlp[0].weight), f32(blk.mlp[0].bias)
            b.fc2_w, b.fc2_b = shadow(blk.mlp[2].weight), f32(blk.mlp[2].bias)
            b.ada0_w, b.ada0_b = shadow(blk.adaLN_modulation[0].weight, out=ada0_all[i]), f32(blk.adaLN_modulation[0].bias)
            b.ada2_w, b.ada2_b = shadow(blk.adaLN_modulation[2].weight, out=ada2_all[i]), f32(blk.adaLN_modulation[2].bias)
        w = N.DitWeights()
        w.enc_w, w.enc_b = shadow(self.dit.patch_encoder.weight, kpad), f32(self.dit.patch_encoder.bias)
        w.pos = f32(self.dit.patch_pos_embedding)
        w.t_scale, w.t_bias = f32(self.dit.t_embedding.scale), f32(self.dit.t_embedding.bias)
        w.dec_ln_w, w.dec_ln_b = f32(self.dit.patch_decoder[0].weight), f32(self.dit.patch_decoder[0].bias)
        w.dec_w, w.dec_b = f32(self.dit.patch_decoder[1].weight), f32(self.dit.patch_decoder[1].bias)
        w.blocks = C.cast(blocks, C.POINTER(N.DitBlockWeights))
        plan = {"skey": skey, "pack": (cfg, w, blocks, keep), "descs": descs, "table": None, "pack_t": None}
        self._finish_table(plan)
        return plan

    def _finish_table(self, plan):
        """(Re)build the device descriptor table from plan["descs"] (rows of [src, dst, dst_t, rows, cols, ld, ld_t])."""
        lib = N.lib()
        arr = (N.CastDesc * len(plan["descs"]))()
        tiles = 0
        for i, (src, dst, dst_t, rows, cols, ld, ld_t) in enumerate(plan["descs"]):
            arr[i] = N.CastDesc(src.data_ptr(), dst.data_ptr() if dst is not None else None,
                                dst_t.data_ptr() if dst_t is not None else None, rows, cols, ld, ld_t, tiles, 0)
            tiles += lib.bsi_cast_batch_tiles(rows, cols, ld if dst is not None else cols)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        plan["table"] = (raw.to(plan["descs"][0][0].device), len(plan["descs"]), tiles)
        # The same descriptors grouped by the GATES of a gated training forward (bsi_dit_train_forward_set_gates): [0] the patch
        # encoder (the last descriptor), [1 + l] the six matrices of block l, [depth + 1] nothing -- one device table of sub-tables
        # whose tile0 restart at 0, and per gate (first descriptor, count, tiles).
        depth = len(self.dit.blocks)
        if len(plan["descs"]) == 6 * depth + 1:
            groups = [[6 * depth]] + [list(range(6 * l, 6 * l + 6)) for l in range(depth)] + [[]]
            sub = (N.CastDesc * len(plan["descs"]))()
            spans, pos = [], 0
            for idx in groups:
                t0 = 0
                for j in idx:
                    src, dst, dst_t, rows, cols, ld, ld_t = plan["descs"][j]
                    sub[pos + len([k for k in idx if k < j])] = N.CastDesc(
                        src.data_ptr(), dst.data_ptr() if dst is not None else None, dst_t.data_ptr() if dst_t is not None else None,
                        rows, cols, ld, ld_t, t0, 0)
                    t0 += lib.bsi_cast_batch_tiles(rows, cols, ld if dst is not None else cols)
                spans.append((pos, len(idx), t0))
                pos += len(idx)
            raw_g = torch.frombuffer(bytearray(bytes(sub)), dtype=torch.uint8).to(plan["descs"][0][0].device)
            plan["gate_tables"] = (raw_g, spans)
        else:
            plan["gate_tables"] = None

    def native_pack_uncast(self):
        """The pack WITHOUT refreshing the shadows: for a caller that casts them itself, part by part, as the parameters arrive (the
        gated training forward of a sharded data-parallel step, DPTrainer).  Marks the shadows as current for this parameter
        version -- the caller's casts are enqueued in front of every use."""
        key = self._weights_key()
        skey = self._storage_key()
        if getattr(self, "_plan", None) is None or self._plan["skey"] != skey:
            self._plan = self._build_plan(skey)
        plan = self._plan
        self._pack, self._pack_key = plan["pack"], key
        if plan["pack_t"] is not None:
            self._pack_t, self._pack_t_key = plan["pack_t"], key
        return plan

    def native_pack(self):
        """bf16 [N][K] shadows of the GEMM weights + the ctypes weight table.  The shadows are PERSISTENT buffers described once by a
        plan (rebuilt only when the parameters' storage moves); a new parameter version -- every optimizer step in training --
        refreshes all of them, and the transposed ones of the training path when they exist, with ONE launch
        (`bsi_cast_batch_bf16`; before round 4: 145 + 120 launches of 8 us and as many allocations per step)."""
        key = self._weights_key()
        if self._pack is not None and self._pack_key == key:
            return self._pack
        sync = self.__dict__.get("_params_pending_sync")
        if sync is not None:  # a trainer's all-gather of these parameters is still in flight on another stream: wait for it first
            sync()
        skey = self._storage_key()
        if getattr(self, "_plan", None) is None or self._plan["skey"] != skey:
            self._plan = self._build_plan(skey)
        plan = self._plan
        table, n, tiles = plan["table"]
        with torch.no_grad():
            N.check(N.lib().bsi_cast_batch_bf16(N.ptr(table), n, tiles, N.stream()))
        self._pack, self._pack_key = plan["pack"], key
        if plan["pack_t"] is not None:  # the table carries the transposed shadows too
            self._pack_t, self._pack_t_key = plan["pack_t"], key
        return self._pack

    def _workspace(self, nbytes: int, dev):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._ws

    def adaln_table(self, t: Tensor) -> Tensor:
        """adaLN modulation rows for the times `t` ([R] fp32): returns fp32 [R, depth, 6*dim]
        (dit.py:77-81,90-92 for every block).  Depends only on t, so `BSI.sample` builds it once for the
        whole schedule."""
        cfg, w, _, _ = self.native_pack()
        lib = N.lib()
        t = t.detach().to(torch.float32).contiguous()
        R = t.numel()
        mod = torch.empty((R, cfg.depth, 6 * cfg.dim), dtype=torch.float32, device=t.device)
        scratch = torch.empty(lib.bsi_dit_adaln_scratch_bytes(C.byref(cfg), R), dtype=torch.uint8, device=t.device)
        N.check(lib.bsi_dit_adaln(C.byref(cfg), C.byref(w), N.ptr(t), R, N.ptr(mod), N.ptr(scratch), N.stream()))
        return mod

    def forward_native(self, mu: Tensor, mod: Tensor, *, c_in=None, c_skip=None, c_out=None, coef_stride=1,
                       out: Tensor | None = None, return_tokens: bool = False):
        """One engine call: out = c_skip*mu + c_out*f(c_in*mu) (or f(mu) without coefficients).
        `mod`: [1 or B, depth, 6*dim] rows of `adaln_table`."""
        cfg, w, _, _ = self.native_pack()
        lib = N.lib()
        if mu.dtype != torch.float32:
            raise RuntimeError("bsi_amd.DenoisingDiT: input must be fp32 (bf16 is used inside the kernels)")
        mu = mu.contiguous()
        B = mu.shape[0]
        assert tuple(mu.shape[1:]) == self.data_shape, f"expected [B,{self.data_shape}], got {tuple(mu.shape)}"
        if out is None:
            out = torch.empty_like(mu)
        ws = self._workspace(lib.bsi_dit_workspace_bytes(C.byref(cfg), B), mu.device)
        tokens = None
        if return_tokens:
            tokens = torch.empty((B * lib.bsi_dit_tokens(C.byref(cfg)), cfg.dim), dtype=torch.float32, device=mu.device)
        pair = _pair_default() if self.cu_pair == "env" else self.cu_pair
        if pair is not None and not return_tokens and B >= PAIR_MIN_BATCH and not torch.cuda.is_current_stream_capturing():
            # two half-batch chains over a G/H pair of CU-masked streams; bit-identical to the one-stream call
            N.check(lib.bsi_dit_forward_pair(C.byref(cfg), C.byref(w), B, N.ptr(mu), N.ptr(mod), mod.shape[0],
                                             N.ptr(c_in), N.ptr(c_skip), N.ptr(c_out), coef_stride, N.ptr(out), N.ptr(ws),
                                             cu_pair_handle(mu.device, pair[0]), pair[1], N.stream()))
        else:
            N.check(lib.bsi_dit_forward(C.byref(cfg), C.byref(w), B, N.ptr(mu), N.ptr(mod), mod.shape[0],
                                        N.ptr(c_in), N.ptr(c_skip), N.ptr(c_out), coef_stride, N.ptr(out), N.ptr(ws),
                                        N.ptr(tokens), N.stream()))
        return (out, tokens) if return_tokens else out

    def forward_train(self, mu: Tensor, t: Tensor, c_in=None, c_skip=None, c_out=None) -> Tensor:
        """Differentiable evaluation (parameters only): tape-recording HIP forward + hand-written HIP backward."""
        from .dit_train import dit_forward_train
        return dit_forward_train(self, mu, t, c_in, c_skip, c_out)

    def forward(self, mu: Tensor, t: Tensor) -> Tensor:
        """f(mu, t): mu [B, *data_shape] fp32, t [B] in [0, 1]  (dit.py:225-233)."""
        if not mu.is_cuda:
            raise RuntimeError("bsi_amd.DenoisingDiT: input is not on a HIP device; there is no CPU path")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .dit_train import dit_forward_autograd  # backward kernels live in the training engine
            return dit_forward_autograd(self, mu, t)
        mod = self.adaln_table(t)
        return self.forward_native(mu, mod)
