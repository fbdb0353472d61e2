"""Mirror of bsi/models/vdm_unet.py:20-100 (DenoisingVDMUNet) with bsi/nn/{residual_block,attention,simplified_unet,
sequential}.py on the native UNet engine (`bsi_unet_film` + `bsi_unet_forward`, include/bsi_hip.h).

Module tree, constructor signature and state-dict keys are those of the reference: the torch
submodules only HOLD the fp32 parameters; the forward is implicit-GEMM convolutions on bf16 MFMA with fused
GroupNorm/SiLU/FiLM/residual epilogues, fused attention and an fp32 feature map between blocks."""
import ctypes as C

import torch
from torch import Tensor, nn

from .. import _native as N
from ..nn import FourierFeatures
from .pos_emb import NyquistPositionalEmbedding
from .utils import actfn_from_str


class _Slot(nn.Identity):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices (FeatureModulation, ActFn)."""


class Attention2D(nn.Module):
    """Parameter holder for bsi/nn/attention.py:21-41 (to_qkv / to_out 3x3 convolutions)."""

    def __init__(self, dim: int, *, heads: int = 4, padding_mode: str = "zeros"):
        super().__init__()
        self.heads = heads
        self.to_qkv = nn.Conv2d(dim, dim * 3, 3, padding=1, padding_mode=padding_mode)
        self.to_out = nn.Conv2d(dim, dim, 3, padding=1, padding_mode=padding_mode)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class ResidualBlock(nn.Module):
    """Parameter holder for bsi/nn/residual_block.py:27-64."""

    def __init__(self, dim_in, dim_out, *, c_dim: int, ActFn, Norm, dropout, attention: bool = True,
                 padding_mode: str = "zeros"):
        super().__init__()
        if attention:
            raise NotImplementedError("bsi_amd.ResidualBlock: downsampling_attention=True is not on the native path "
                                      "(no experiment of the reference uses it)")
        self.project_onto_scale_shift = nn.Linear(c_dim, dim_out * 2, 1)
        self.skip = nn.Conv2d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()
        self.layers = nn.Sequential(
            Norm(dim_in), ActFn(), nn.Conv2d(dim_in, dim_out, 3, padding=1, padding_mode=padding_mode), _Slot(), ActFn(),
            *([nn.Dropout(dropout)] if dropout is not None else []),
            nn.Conv2d(dim_out, dim_out, 3, padding=1, padding_mode=padding_mode))
        self.attention = attention
        self.res_attention = nn.Identity()


class SimplifiedUNet(nn.Module):
    """Parameter holder for bsi/nn/simplified_unet.py:6-48."""

    def __init__(self, downsampling_blocks, upsampling_blocks, center_block):
        super().__init__()
        assert len(downsampling_blocks) == len(upsampling_blocks)
        self.downsampling_blocks = nn.ModuleList([nn.ModuleList([b]) for b in downsampling_blocks])
        self.upsampling_blocks = nn.ModuleList([nn.ModuleList([b]) for b in upsampling_blocks])
        self.center_block = center_block


class DenoisingVDMUNet(nn.Module):
    """U-Net as in the VDM paper without down-sampling — same constructor as bsi.models.vdm_unet.DenoisingVDMUNet."""

    def __init__(self, data_shape, pos_emb: NyquistPositionalEmbedding, actfn: str, dim: int, levels: int,
                 pos_emb_mult: int, n_attention_heads: int = 1, dropout: float | None = None,
                 downsampling_attention: bool = False, fourier_features: FourierFeatures | None = None,
                 padding_mode: str = "zeros", **kwargs):
        super().__init__()
        self.data_shape = tuple(data_shape)
        self.pos_emb = pos_emb
        self.fourier_features = fourier_features
        assert len(self.data_shape) == 3, "Only works for 2D images"
        if actfn != "silu" or padding_mode != "zeros":
            raise NotImplementedError("bsi_amd.DenoisingVDMUNet: the native kernels implement actfn='silu' and "
                                      "padding_mode='zeros' (the reference's configuration)")
        n_channels = data_shape[0]
        in_features = out_features = n_channels
        if fourier_features is not None:
            in_features += n_channels * fourier_features.n_features()
        ActFn = actfn_from_str(actfn)

        def Norm(c):
            return nn.GroupNorm(32, c)

        def residual_block(din, dout, c_dim):
            return ResidualBlock(din, dout, c_dim=c_dim, ActFn=ActFn, Norm=Norm, dropout=dropout,
                                 attention=downsampling_attention, padding_mode=padding_mode)

        c_dim = pos_emb.size * pos_emb_mult
        self.pos_map = nn.Sequential(self.pos_emb, nn.Linear(pos_emb.size, c_dim), ActFn(), nn.Linear(c_dim, c_dim), ActFn())
        self.encode = nn.Conv2d(in_features, dim, 3, padding=1, padding_mode=padding_mode)
        self.decode = nn.Conv2d(dim, out_features, 1)
        down = [residual_block(dim, dim, c_dim) for _ in range(levels)]
        up = [residual_block(2 * dim, dim, c_dim) for _ in range(levels)]
        center = nn.Sequential(residual_block(dim, dim, c_dim),
                               Residual(nn.Sequential(Norm(dim), Attention2D(dim, heads=n_attention_heads,
                                                                             padding_mode=padding_mode))),
                               residual_block(dim, dim, c_dim))
        self.u_net = SimplifiedUNet(down, up, center)
        self._cfg_args = dict(dim=dim, levels=levels, heads=n_attention_heads, c_dim=c_dim)
        self._dropout = dropout
        self._pack = None
        self._pack_key = None
        self._pack_t = None
        self._pack_t_key = None
        self._plan = None
        self._plan_t = None
        self._ws = None

    # ------------------------------------------------------------------------------------------------
    def _config(self) -> N.UNetConfig:
        Cc, H, W = self.data_shape
        ff = self.fourier_features
        a = self._cfg_args
        return N.UNetConfig(Cc, H, W, a["dim"], a["levels"], a["heads"], ff.n_min if ff is not None else 1,
                            ff.n_max if ff is not None else 0, self.pos_emb.size, a["c_dim"])

    _NATIVE_CACHES = ("_pack", "_pack_key", "_pack_t", "_pack_t_key", "_plan", "_plan_t", "_plan_g", "_ws", "_last_flat_grad", "_grad_buffer")

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

    def _blocks(self):
        u = self.u_net
        return ([b[0] for b in u.downsampling_blocks] + [u.center_block[0], u.center_block[2]] +
                [b[0] for b in u.upsampling_blocks])

    def _storage_key(self):
        """Identity of the parameters' STORAGE: the pack plan bakes every parameter's raw pointer into device-side descriptor
        tables, so the key names every one of them (330 integers; a re-pack costs far more) -- replacing the storage of an
        interior parameter (`p.data = ...`, `load_state_dict(assign=True)`, a per-module `.to()`) must rebuild the plan."""
        ps = list(self.parameters())
        return (ps[0].device, len(ps), hash(tuple(p.data_ptr() for p in ps)))

    def native_pack(self):
        """(config, weight table, block array, keep-alive) for the HIP engines.  The bf16 shadows live in PERSISTENT buffers described
        once by a plan (rebuilt only when the parameters' storage moves); a new parameter version -- every optimizer step in
        training -- refreshes them with a handful of launches (one batched re-pack of all convolution weights, concatenations
        and casts of the FiLM / pos_map matrices) instead of ~170 small ones."""
        key = self._weights_key()
        if self._pack is not None and self._pack_key == key:
            return self._pack
        skey = self._storage_key()
        if getattr(self, "_plan", None) is None or self._plan["skey"] != skey:
            self._plan = self._build_plan(skey)
        plan = self._plan
        lib = N.lib()
        with torch.no_grad():
            N.check(lib.bsi_conv_weight_pack_batch(N.ptr(plan["descs"]), plan["ndesc"], 0, N.stream()))
            if plan["b2_dst"] is not None:  # conv2 bias + skip conv bias (both add to the block output, residual_block.py:63)
                torch.stack(plan["b2_a"], out=plan["b2_tmp"])
                torch.stack(plan["b2_b"], out=plan["b2_dst"])
                plan["b2_dst"].add_(plan["b2_tmp"])
            torch.cat(plan["film_ws"], dim=0, out=plan["film_w_cat"])
            torch.cat(plan["film_bs"], dim=0, out=plan["film_b_cat"])
            for src, rows, cols, dst, ld in plan["casts"]:
                N.check(lib.bsi_cast_bf16(N.ptr(src), rows, cols, N.ptr(dst), ld, N.stream()))
        self._pack = plan["pack"]
        self._pack_key = key
        return self._pack

    def _build_plan(self, skey):
        lib = N.lib()
        dev = self.encode.weight.device
        if dev.type != "cuda":
            raise RuntimeError("bsi_amd.DenoisingVDMUNet: parameters must live on a HIP device (no CPU path)")
        cfg = self._config()
        keep, descs, casts = [], [], []

        def f32(p: Tensor):
            t = p.detach()
            assert t.is_contiguous(), "parameters must be contiguous"
            keep.append(t)
            return t.data_ptr()

        def conv_pack(conv: nn.Conv2d, cin_pad=None, extra: nn.Conv2d | None = None):
            cout, cin, kh, kw = conv.weight.shape
            taps = kh * kw
            cin_pad = cin_pad or cin
            k = taps * cin_pad + (extra.weight.shape[1] if extra is not None else 0)
            out = torch.empty((cout, k), dtype=torch.bfloat16, device=dev)
            descs.append(N.ConvPackDesc(f32(conv.weight), out.data_ptr(), cout, cin, taps, cin_pad, k, 0))
            if extra is not None:
                c2 = extra.weight.shape[1]
                descs.append(N.ConvPackDesc(f32(extra.weight), out.data_ptr(), cout, c2, 1, c2, k, taps * cin_pad))
            keep.append(out)
            return out.data_ptr()

        def lin_shadow(w: Tensor, ld=None):
            w = w.detach()
            assert w.is_contiguous()
            rows, cols = w.shape
            ld = ld or cols
            out = torch.zeros((rows, ld), dtype=torch.bfloat16, device=dev)  # padding columns (ld > cols) stay zero
            casts.append((w, rows, cols, out, ld))
            keep.extend([w, out])
            return out.data_ptr()

        blocks = self._blocks()
        dim = cfg.dim
        skips = [i for i, rb in enumerate(blocks) if isinstance(rb.skip, nn.Conv2d)]
        b2_dst = torch.empty((len(skips), dim), dtype=torch.float32, device=dev) if skips else None
        b2_tmp = torch.empty_like(b2_dst) if skips else None
        arr = (N.UNetResBlockWeights * len(blocks))()
        for i, rb in enumerate(blocks):
            conv2 = rb.layers[-1]
            has_skip = isinstance(rb.skip, nn.Conv2d)
            arr[i].gn_w, arr[i].gn_b = f32(rb.layers[0].weight), f32(rb.layers[0].bias)
            arr[i].conv1_w, arr[i].conv1_b = conv_pack(rb.layers[2]), f32(rb.layers[2].bias)
            arr[i].conv2_w = conv_pack(conv2, extra=rb.skip if has_skip else None)
            arr[i].conv2_b = b2_dst[skips.index(i)].data_ptr() if has_skip else f32(conv2.bias)
        w = N.UNetWeights()
        w.enc_w, w.enc_b = conv_pack(self.encode, cin_pad=lib.bsi_unet_cin_pad(C.byref(cfg))), f32(self.encode.bias)
        w.dec_w, w.dec_b = f32(self.decode.weight.detach().reshape(self.decode.weight.shape[0], -1)), f32(self.decode.bias)
        w.pe_scale, w.pe_bias = f32(self.pos_emb.scale), f32(self.pos_emb.bias)
        w.pm1_w, w.pm1_b = lin_shadow(self.pos_map[1].weight, 64), f32(self.pos_map[1].bias)
        w.pm3_w, w.pm3_b = lin_shadow(self.pos_map[3].weight), f32(self.pos_map[3].bias)
        film_ws = [rb.project_onto_scale_shift.weight.detach() for rb in blocks]
        film_bs = [rb.project_onto_scale_shift.bias.detach() for rb in blocks]
        film_w_cat = torch.empty((sum(t.shape[0] for t in film_ws), film_ws[0].shape[1]), dtype=torch.float32, device=dev)
        film_b_cat = torch.empty(film_w_cat.shape[0], dtype=torch.float32, device=dev)
        w.film_w, w.film_b = lin_shadow(film_w_cat), film_b_cat.data_ptr()
        att = self.u_net.center_block[1].fn
        w.agn_w, w.agn_b = f32(att[0].weight), f32(att[0].bias)
        w.aqkv_w, w.aqkv_b = conv_pack(att[1].to_qkv), f32(att[1].to_qkv.bias)
        w.aout_w, w.aout_b = conv_pack(att[1].to_out), f32(att[1].to_out.bias)
        w.blocks = C.cast(arr, C.POINTER(N.UNetResBlockWeights))
        darr = (N.ConvPackDesc * len(descs))(*descs)
        descs_dev = torch.frombuffer(bytearray(bytes(darr)), dtype=torch.uint8).to(dev)
        keep.extend([b2_dst, b2_tmp, film_w_cat, film_b_cat, descs_dev, film_ws, film_bs])
        return {"skey": skey, "pack": (cfg, w, arr, keep), "descs": descs_dev, "ndesc": len(descs), "casts": casts,
                "b2_dst": b2_dst, "b2_tmp": b2_tmp, "b2_a": [blocks[i].layers[-1].bias.detach() for i in skips],
                "b2_b": [blocks[i].skip.bias.detach() for i in skips], "film_ws": film_ws, "film_bs": film_bs,
                "film_w_cat": film_w_cat, "film_b_cat": film_b_cat}

    def _workspace(self, nbytes: int, dev):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._ws

    def adaln_table(self, t: Tensor) -> Tensor:
        """Conditioning table for the times `t` ([R]): FiLM (scale, shift) of every residual block,
        fp32 [R, 2*levels+2, 2*dim] (pos_map of vdm_unet.py:62-69 + project_onto_scale_shift of residual_block.py:39,62).
        Named like the DiT's table so that `BSI.sample` builds it once for the whole schedule."""
        cfg, w, arr, _ = self.native_pack()
        lib = N.lib()
        t = t.detach().to(torch.float32).contiguous()
        R = t.numel()
        film = torch.empty((R, len(arr), 2 * cfg.dim), dtype=torch.float32, device=t.device)
        scratch = torch.empty(lib.bsi_unet_film_scratch_bytes(C.byref(cfg), R), dtype=torch.uint8, device=t.device)
        N.check(lib.bsi_unet_film(C.byref(cfg), C.byref(w), N.ptr(t), R, N.ptr(film), N.ptr(scratch), N.stream()))
        return film

    def forward_native(self, mu: Tensor, mod: Tensor, *, c_in=None, c_skip=None, c_out=None, coef_stride=1,
                       out: Tensor | None = None):
        cfg, w, _, _ = self.native_pack()
        lib = N.lib()
        if mu.dtype != torch.float32:
            raise RuntimeError("bsi_amd.DenoisingVDMUNet: input must be fp32 (bf16 is used inside the kernels)")
        mu = mu.contiguous()
        B = mu.shape[0]
        assert tuple(mu.shape[1:]) == self.data_shape, f"expected [B,{self.data_shape}], got {tuple(mu.shape)}"
        if out is None:
            out = torch.empty_like(mu)
        ws = self._workspace(lib.bsi_unet_workspace_bytes(C.byref(cfg), B), mu.device)
        N.check(lib.bsi_unet_forward(C.byref(cfg), C.byref(w), B, N.ptr(mu), N.ptr(mod), mod.shape[0], N.ptr(c_in),
                                     N.ptr(c_skip), N.ptr(c_out), coef_stride, N.ptr(out), N.ptr(ws), N.stream()))
        return out

    def forward(self, mu: Tensor, t: Tensor) -> Tensor:
        """f(mu, t): mu [B, *data_shape] fp32, t [B] in [0, 1]  (vdm_unet.py:92-100)."""
        if not mu.is_cuda:
            raise RuntimeError("bsi_amd.DenoisingVDMUNet: input is not on a HIP device; there is no CPU path")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self.forward_train(mu, t)  # backward kernels live in the training engine
        return self.forward_native(mu, self.adaln_table(t))

    def forward_train(self, mu: Tensor, t: Tensor, c_in=None, c_skip=None, c_out=None) -> Tensor:
        """Differentiable evaluation (parameters only): tape-recording HIP forward + hand-written HIP backward."""
        from .unet_train import unet_forward_train
        return unet_forward_train(self, mu, t, c_in, c_skip, c_out)
