"""Mirror of bsi/vdm.py of the reference (Variational Diffusion Models, arXiv 2107.00630) on the native kernels of
include/bsi_hip.h (`bsi_vdm_*`, `bsi_affine_noise`, `bsi_axpbypcz`, `bsi_sqerr_rows`, ...) and the same denoiser engines
as `bsi_amd.BSI`.  Same constructor, attributes, methods, return shapes and RNG draw order as `bsi.vdm.VDM`; Python is
glue (noise draws from the caller's generator, allocation, kernel enqueues)."""
import math

import torch
from torch import Tensor, nn

from . import _native as N
from .bsi import Discretization, _new, _PredictCombine, _SqErr


class VDM(nn.Module):
    """Drop-in for `bsi.vdm.VDM` (vdm.py:13-401).  The denoiser predicts the noise: x_hat = (z_t - sigma_t f(z_t, t)) / alpha_t."""

    def __init__(self, model: nn.Module, *, data_shape: tuple[int, ...], snr_min: float, snr_max: float, k: int,
                 low_discrepancy_sampling: bool = True, discretization: Discretization | None = None):
        super().__init__()
        self._model = [model]  # not a submodule, not in the state dict (vdm.py:32-34)
        self.data_shape = tuple(data_shape)
        self.snr_min = snr_min
        self.snr_max = snr_max
        self.k = k
        self.low_discrepancy_sampling = low_discrepancy_sampling
        self.discretization = discretization
        self.register_buffer("_gamma_0", -torch.as_tensor(snr_max).log(), persistent=False)
        self.register_buffer("_gamma_1", -torch.as_tensor(snr_min).log(), persistent=False)
        self._g0, self._g1 = float(self._gamma_0), float(self._gamma_1)  # fp32 values of the buffers at construction
        self._D = math.prod(self.data_shape)

    @property
    def model(self):
        return self._model[0]

    def set_model(self, model):
        self._model[0] = model

    @property
    def tensor_args(self):
        return {"device": self._gamma_0.device, "dtype": self._gamma_0.dtype}

    # -- helpers ---------------------------------------------------------------------------------------
    def _require(self):
        if self._gamma_0.dtype != torch.float32:
            raise RuntimeError(f"bsi_amd.VDM: the native path computes the wrapper in fp32 (got {self._gamma_0.dtype})")
        if self._gamma_0.device.type != "cuda":
            raise RuntimeError("bsi_amd.VDM: module is not on a HIP device; there is no CPU path")

    def _coeffs(self, t: Tensor, which=("alpha", "sigma", "snr", "c_skip", "c_out")):
        self._require()
        t = t.to(torch.float32).contiguous()
        n = t.numel()
        out = {k: _new(t.shape, t) for k in which}
        N.check(N.lib().bsi_vdm_coeffs(N.ptr(t), n, self._g0, self._g1, N.ptr(out.get("alpha")), N.ptr(out.get("sigma")),
                                       N.ptr(out.get("snr")), N.ptr(out.get("c_skip")), N.ptr(out.get("c_out")), N.stream()))
        return out

    def _native_model(self):
        m = self.model
        return m if hasattr(m, "forward_native") and hasattr(m, "adaln_table") else None

    # -- schedule functions (vdm.py:138-150) ------------------------------------------------------------
    def gamma(self, t: Tensor) -> Tensor:
        return torch.lerp(self._gamma_0, self._gamma_1, t)

    def sigma2(self, t: Tensor) -> Tensor:
        s = self._coeffs(t, ("sigma",))["sigma"]
        return s * s

    def alpha(self, t: Tensor) -> Tensor:
        return self._coeffs(t, ("alpha",))["alpha"]

    def snr(self, t: Tensor) -> Tensor:
        return self._coeffs(t, ("snr",))["snr"]

    # -- ELBO (vdm.py:60-136) ----------------------------------------------------------------------------
    def _assemble(self, l_prior, l_recon, l_diff, n_recon_samples, n_measure_samples, estimate_var):
        elbo = -(l_prior + l_recon.mean(dim=0) + l_diff.mean(dim=0))
        conversion_factor = -1 / (math.log(2) * math.prod(self.data_shape))
        bpd = conversion_factor * elbo
        extra = {"l_prior": l_prior, "l_recon": l_recon, "l_diff": l_diff}
        if estimate_var:
            assert n_recon_samples > 1 and n_measure_samples > 1, (
                "Need at least two samples of each to estimate variance")
            l_recon_var = l_recon.var(dim=0, unbiased=True) / n_recon_samples
            l_diff_var = l_diff.var(dim=0, unbiased=True) / n_measure_samples
            extra["bpd_var"] = (conversion_factor**2) * (l_recon_var + l_diff_var)
        return elbo, bpd, extra

    def elbo(self, x: Tensor, n_recon_samples: int, n_measure_samples: int, generator=None, *, estimate_var: bool = False):
        l_prior = self.prior_loss(x)
        l_recon = self.reconstruction_loss(x, n_recon_samples, generator)
        l_diff = self.inf_diffusion_loss(x, n_measure_samples, generator)
        return self._assemble(l_prior, l_recon, l_diff, n_recon_samples, n_measure_samples, estimate_var)

    def finite_elbo(self, x: Tensor, n_recon_samples: int, n_measure_samples: int, generator=None, *,
                    t: Tensor | None = None, estimate_var: bool = False):
        l_prior = self.prior_loss(x)
        l_recon = self.reconstruction_loss(x, n_recon_samples, generator)
        l_diff = self.finite_diffusion_loss(x, n_measure_samples, generator, t=t)
        return self._assemble(l_prior, l_recon, l_diff, n_recon_samples, n_measure_samples, estimate_var)

    def prior_loss(self, x: Tensor) -> Tensor:
        """vdm.py:127-136 -> [B]."""
        self._require()
        x = x.contiguous()
        sig1 = float(self._coeffs(x.new_ones((1,)), ("sigma",))["sigma"])
        out = _new((len(x),), x)
        N.check(N.lib().bsi_vdm_prior(N.ptr(x), sig1 * sig1, len(x), self._D, N.ptr(out), N.stream()))
        return out

    def reconstruction_loss(self, x: Tensor, n_samples: int, generator=None) -> Tensor:
        """vdm.py:152-195 -> [n_samples, B]."""
        self._require()
        lib = N.lib()
        x = x.contiguous()
        B, rows = len(x), n_samples * len(x)
        c0 = self._coeffs(x.new_zeros((1,)), ("alpha", "sigma", "c_skip"))
        eps = torch.randn((n_samples, *x.shape), device=x.device, dtype=x.dtype, generator=generator)
        z0 = torch.empty_like(eps)
        a_rows, b_rows = c0["alpha"].expand(rows).contiguous(), c0["sigma"].expand(rows).contiguous()  # kept alive past the launch
        N.check(lib.bsi_affine_noise(N.ptr(x), N.ptr(a_rows), N.ptr(b_rows), N.ptr(eps), rows, B, self._D, N.ptr(z0), N.stream()))
        x_hat = torch.empty_like(z0)
        N.check(lib.bsi_scale_rows(N.ptr(z0), N.ptr(c0["c_skip"]), 0, rows, self._D, N.ptr(x_hat), N.stream()))  # z_0 / alpha_0
        std = float(c0["sigma"]) / float(c0["alpha"])
        out = _new((rows,), x)
        d = self.discretization
        if d is None:
            # -log N(x; x_hat, std): the continuous branch of bsi_recon_nll with alpha_R = 1/std^2
            N.check(lib.bsi_recon_nll(N.ptr(x), N.ptr(x_hat), 1.0 / (std * std), None, 0.0, 1.0, 0, rows, B, self._D, N.ptr(out),
                                      N.stream()))
        else:
            bounds = d.bin_boundaries(x.device, x.dtype).contiguous()
            N.check(lib.bsi_vdm_recon_nll(N.ptr(x), N.ptr(x_hat), std, N.ptr(bounds), d.min - d.dx / 2, d.dx, d.k, rows, B, self._D,
                                          N.ptr(out), N.stream()))
        return out.reshape(n_samples, B)

    def diffusion_loss(self, x: Tensor, n_samples: int, generator=None) -> Tensor:
        raise NotImplementedError()

    def finite_diffusion_loss(self, x: Tensor, n_samples: int, generator=None, *, t: Tensor | None = None) -> Tensor:
        """vdm.py:206-231 -> [n_samples, B]."""
        self._require()
        if t is None:
            t = torch.linspace(1.0, 0.0, self.k + 1, **self.tensor_args)
        t = t.to(torch.float32).contiguous()
        x = x.contiguous()
        T = len(t) - 1
        B = len(x)
        i = torch.randint(0, T, (n_samples, B), device=x.device, generator=generator)
        cz, cx, sd, dsnr = (_new((T,), t) for _ in range(4))
        N.check(N.lib().bsi_vdm_step_coeffs(N.ptr(t), T, self._g0, self._g1, N.ptr(cz), N.ptr(cx), N.ptr(sd), N.ptr(dsnr), N.stream()))
        t_i = t[i]
        z_t = self._sample_zt_given_x(x, t_i, generator)
        x_hat = self._predict_x(z_t.flatten(end_dim=1), t_i.flatten(end_dim=1))
        w = dsnr[i].flatten().contiguous()  # snr(s_i) - snr(t_i)
        return _SqErr.apply(x, x_hat, w, 0.5 * T, False).reshape(n_samples, B)

    def inf_diffusion_loss(self, x: Tensor, n_samples: int, generator=None, *, _scale: float = 1.0) -> Tensor:
        """vdm.py:233-249 -> [n_samples, B]."""
        self._require()
        x = x.contiguous()
        B = len(x)
        t = self._sample_t(n_samples, B, generator)
        z_t = self._sample_zt_given_x(x, t, generator)
        tf = t.flatten().contiguous()
        x_hat = self._predict_x(z_t.flatten(end_dim=1), tf)
        snr = self._coeffs(tf, ("snr",))["snr"]
        # dsnr/dt = -snr(t) * (gamma_0 - gamma_1): gamma is linear in t
        scale = 0.5 * -(self._g0 - self._g1) * _scale
        return _SqErr.apply(x, x_hat, snr, scale, False).reshape(t.shape[0], -1)

    def train_loss(self, x: Tensor, generator=None) -> Tensor:
        """One sample of the infinite-step diffusion loss with a mean over the data dimensions (vdm.py:251-262) -> [1, B]."""
        return self.inf_diffusion_loss(x, 1, generator, _scale=1.0 / math.prod(self.data_shape))

    # -- sampling (vdm.py:264-322) -------------------------------------------------------------------------
    def _chain(self, n_samples, generator, t, history):
        self._require()
        lib = N.lib()
        ts = torch.linspace(1.0, 0.0, self.k + 1, **self.tensor_args) if t is None else t.to(torch.float32).contiguous()
        k = ts.numel() - 1
        shape = (n_samples, *self.data_shape)
        cz, cx, sd = (_new((k,), ts) for _ in range(3))
        N.check(lib.bsi_vdm_step_coeffs(N.ptr(ts), k, self._g0, self._g1, N.ptr(cz), N.ptr(cx), N.ptr(sd), None, N.stream()))
        co = self._coeffs(ts, ("c_skip", "c_out"))
        ones = torch.ones(k + 1, **self.tensor_args)
        native = self._native_model()
        mod = native.adaln_table(ts) if native is not None else None
        if history:
            x_hats = torch.zeros((self.k + 1, *shape), **self.tensor_args)
        z = torch.randn(shape, **self.tensor_args, generator=generator)
        z_next = torch.empty_like(z)
        x_hat = torch.empty_like(z)
        n_elem = z.numel()
        for i in range(k):
            if native is not None:
                native.forward_native(z, mod[i:i + 1], c_in=ones[i:], c_skip=co["c_skip"][i:], c_out=co["c_out"][i:], coef_stride=0,
                                      out=x_hat)
            else:
                f = self.model(z, ts[i].clone().repeat(n_samples)).contiguous()
                N.check(lib.bsi_predict_combine(N.ptr(z), N.ptr(f), N.ptr(co["c_skip"][i:]), N.ptr(co["c_out"][i:]), 0, n_samples,
                                                self._D, N.ptr(x_hat), N.stream()))
            if history:
                x_hats[i].copy_(x_hat)
            eps = torch.randn(shape, **self.tensor_args, generator=generator)
            N.check(lib.bsi_axpbypcz(N.ptr(z), N.ptr(x_hat), N.ptr(eps), N.ptr(cz), N.ptr(cx), N.ptr(sd), i, n_elem, N.ptr(z_next),
                                     N.stream()))
            z, z_next = z_next, z
        c0 = self._coeffs(z.new_zeros((1,)), ("c_skip",))["c_skip"]
        out = torch.empty_like(z)
        N.check(lib.bsi_scale_rows(N.ptr(z), N.ptr(c0), 0, n_samples, self._D, N.ptr(out), N.stream()))  # z / alpha_0
        if history:
            x_hats[-1].copy_(out)
            return x_hats
        return out

    def sample(self, n_samples: int, generator=None, *, t: Tensor | None = None) -> Tensor:
        return self._chain(n_samples, generator, t, history=False)

    def sample_history(self, n_samples: int, generator=None, *, t: Tensor | None = None) -> Tensor:
        return self._chain(n_samples, generator, t, history=True)

    # -- pieces (vdm.py:324-401) ---------------------------------------------------------------------------
    def _predict_x(self, z_t: Tensor, t: Tensor) -> Tensor:
        self._require()
        z_t = z_t.contiguous()
        t = t.to(torch.float32).contiguous()
        co = self._coeffs(t, ("c_skip", "c_out"))
        native = self._native_model()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters())
        if native is not None:
            ones = torch.ones_like(t)
            if not needs_grad:
                return native.forward_native(z_t, native.adaln_table(t), c_in=ones, c_skip=co["c_skip"], c_out=co["c_out"],
                                             coef_stride=1)
            if hasattr(native, "forward_train"):
                return native.forward_train(z_t, t, ones, co["c_skip"], co["c_out"])
        return _PredictCombine.apply(z_t, self.model(z_t, t), co["c_skip"], co["c_out"])

    def _sample_zt_given_x(self, x: Tensor, t: Tensor, generator=None) -> Tensor:
        self._require()
        x = x.contiguous()
        t = t.to(torch.float32).contiguous()
        co = self._coeffs(t.flatten(), ("alpha", "sigma"))
        eps = torch.randn((*t.shape, *self.data_shape), dtype=x.dtype, device=x.device, generator=generator)
        out = torch.empty_like(eps)
        N.check(N.lib().bsi_affine_noise(N.ptr(x), N.ptr(co["alpha"]), N.ptr(co["sigma"]), N.ptr(eps), t.numel(), len(x), self._D,
                                         N.ptr(out), N.stream()))
        return out

    def _sample_zs_given_zt_x(self, s: Tensor, z_t: Tensor, t: Tensor, x: Tensor, generator=None) -> Tensor:
        """Ancestral step for per-sample (s, t) (vdm.py:350-379); `sample` uses the schedule-indexed kernel instead."""
        self._require()
        lib = N.lib()
        n = s.numel()
        out = torch.empty_like(z_t)
        eps = torch.randn(z_t.shape, dtype=x.dtype, device=x.device, generator=generator)
        for r in range(n):  # rows may have different (s, t): one coefficient triple per row
            ts = torch.stack([t[r], s[r]]).to(torch.float32).contiguous()
            cz, cx, sd = (_new((1,), ts) for _ in range(3))
            N.check(lib.bsi_vdm_step_coeffs(N.ptr(ts), 1, self._g0, self._g1, N.ptr(cz), N.ptr(cx), N.ptr(sd), None, N.stream()))
            N.check(lib.bsi_axpbypcz(N.ptr(z_t[r]), N.ptr(x[r].contiguous()), N.ptr(eps[r]), N.ptr(cz), N.ptr(cx), N.ptr(sd), 0,
                                     self._D, N.ptr(out[r]), N.stream()))
        return out

    def _sample_t(self, n_samples: int, batch_size: int, generator=None) -> Tensor:
        self._require()
        if self.low_discrepancy_sampling:
            offset = torch.rand((), **self.tensor_args, generator=generator)
            total = n_samples * batch_size
            perm = torch.randperm(total, device=self.tensor_args["device"], generator=generator)
            t = torch.empty((n_samples, batch_size), **self.tensor_args)
            N.check(N.lib().bsi_tgrid(N.ptr(perm), N.ptr(offset), total, N.ptr(t), N.stream()))
            return t
        return torch.rand((batch_size, n_samples), **self.tensor_args, generator=generator)  # reference's (B, n) shape quirk
