"""Mirror of bsi/bfn.py of the reference (Bayesian Flow Networks, arXiv 2308.07037) on the native kernels of
include/bsi_hip.h (`bsi_bfn_*`, `bsi_affine_noise`, `bsi_clip`, `bsi_refine_step`, `bsi_sqerr_rows`, `bsi_recon_nll`) and the
same denoiser engines as `bsi_amd.BSI`.  Same constructor, attributes, methods, return shapes and RNG draw order as
`bsi.bfn.BFN`; Python is glue (noise draws from the caller's generator, allocation, kernel enqueues)."""
import math

import torch
from torch import Tensor, nn

from . import _native as N
from .bsi import Discretization, _new, _PredictCombine, _SqErr


class _Clip(torch.autograd.Function):
    """x.clip(lo, hi) (bfn.py:291) with torch.clamp's gradient (passes where lo <= x <= hi)."""

    @staticmethod
    def forward(ctx, raw, lo, hi):
        raw = raw.contiguous()
        out = torch.empty_like(raw)
        N.check(N.lib().bsi_clip(N.ptr(raw), lo, hi, raw.numel(), N.ptr(out), N.stream()))
        ctx.save_for_backward(raw)
        ctx.bounds = (lo, hi)
        return out

    @staticmethod
    def backward(ctx, g):
        (raw,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        N.check(N.lib().bsi_clip_bwd(N.ptr(g), N.ptr(raw), ctx.bounds[0], ctx.bounds[1], g.numel(), N.ptr(out), N.stream()))
        return out, None, None


class BFN(nn.Module):
    """Drop-in for `bsi.bfn.BFN` (bfn.py:12-329).  The denoiser predicts the noise:
    x_hat = clip(mu / gamma - sqrt((1 - gamma) / gamma) f(mu, t)), gamma = 1 - sigma_1^(2t)."""

    def __init__(self, model: nn.Module, *, data_shape: tuple[int, ...], sigma_1: float, k: int, x_min: float = -1.0,
                 x_max: float = 1.0, t_min: float = 1e-6, low_discrepancy_sampling: bool = True,
                 discretization: Discretization | None = None):
        super().__init__()
        self._model = [model]  # not a submodule, not in the state dict (bfn.py:33-35)
        self.data_shape = tuple(data_shape)
        assert sigma_1 < 1.0, "`sigma_1 < 1` is required by BFN formulas"
        self.register_buffer("sigma_1", torch.as_tensor(sigma_1), persistent=False)
        self.k = k
        self.x_min = x_min
        self.x_max = x_max
        self.t_min = t_min
        self.low_discrepancy_sampling = low_discrepancy_sampling
        self.discretization = discretization
        self._s1 = float(self.sigma_1)  # fp32 value of the buffer
        self._D = math.prod(self.data_shape)

    @property
    def model(self):
        return self._model[0]

    def set_model(self, model):
        self._model[0] = model

    @property
    def tensor_args(self):
        return {"device": self.sigma_1.device, "dtype": self.sigma_1.dtype}

    # -- helpers ---------------------------------------------------------------------------------------
    def _require(self):
        if self.sigma_1.dtype != torch.float32:
            raise RuntimeError(f"bsi_amd.BFN: the native path computes the wrapper in fp32 (got {self.sigma_1.dtype})")
        if self.sigma_1.device.type != "cuda":
            raise RuntimeError("bsi_amd.BFN: module is not on a HIP device; there is no CPU path")

    def _coeffs(self, t: Tensor, which=("fa", "fb", "c_skip", "c_out", "w")):
        self._require()
        t = t.to(torch.float32).contiguous()
        out = {k: _new(t.shape, t) for k in which}
        N.check(N.lib().bsi_bfn_coeffs(N.ptr(t), t.numel(), self._s1, float(self.t_min), N.ptr(out.get("fa")), N.ptr(out.get("fb")),
                                       N.ptr(out.get("c_skip")), N.ptr(out.get("c_out")), N.ptr(out.get("w")), N.stream()))
        return out

    def _schedule(self, t: Tensor):
        t = t.to(torch.float32).contiguous()
        k = t.numel() - 1
        alpha, rho, wdisc = _new((k,), t), _new((k + 1,), t), _new((k,), t)
        N.check(N.lib().bsi_bfn_schedule(N.ptr(t), k, self._s1, N.ptr(alpha), N.ptr(rho), N.ptr(wdisc), N.stream()))
        return alpha, rho, wdisc

    def _native_model(self):
        m = self.model
        return m if hasattr(m, "forward_native") and hasattr(m, "adaln_table") else None

    # -- ELBO (bfn.py:59-123) ----------------------------------------------------------------------------
    def _assemble(self, l_recon, l_latent, n_recon_samples, n_measure_samples, estimate_var):
        elbo = -(l_recon.mean(dim=0) + l_latent.mean(dim=0))
        conversion_factor = -1 / (math.log(2) * math.prod(self.data_shape))
        bpd = conversion_factor * elbo
        extra = {"l_recon": l_recon, "l_latent": l_latent}
        if estimate_var:
            assert n_recon_samples > 1 and n_measure_samples > 1, (
                "Need at least two samples of each to estimate variance")
            l_recon_var = l_recon.var(dim=0, unbiased=True) / n_recon_samples
            l_latent_var = l_latent.var(dim=0, unbiased=True) / n_measure_samples
            extra["bpd_var"] = (conversion_factor**2) * (l_recon_var + l_latent_var)
        return elbo, bpd, extra

    def elbo(self, x: Tensor, n_recon_samples: int, n_measure_samples: int, generator=None, *, estimate_var: bool = False):
        l_recon = self.reconstruction_loss(x, n_recon_samples, generator)
        l_latent = self.continuous_time_loss(x, n_measure_samples, generator)
        return self._assemble(l_recon, l_latent, n_recon_samples, n_measure_samples, estimate_var)

    def finite_elbo(self, x: Tensor, n_recon_samples: int, n_measure_samples: int, generator=None, *,
                    t: Tensor | None = None, estimate_var: bool = False):
        l_recon = self.reconstruction_loss(x, n_recon_samples, generator)
        l_latent = self.discrete_time_loss(x, n_measure_samples, generator, t=t)
        return self._assemble(l_recon, l_latent, n_recon_samples, n_measure_samples, estimate_var)

    def reconstruction_loss(self, x: Tensor, n_samples: int, generator=None) -> Tensor:
        """bfn.py:125-153 -> [n_samples, B]: Normal(x_hat(t = 1), sigma_1) integrated over the bin of x."""
        self._require()
        x = x.contiguous()
        B, rows = len(x), n_samples * len(x)
        t = x.new_ones((n_samples, B))
        mu = self._sample_flow_distribution(x, t, generator)
        x_hat = self._predict_x(mu.flatten(end_dim=1), t.flatten(end_dim=1)).contiguous()
        out = _new((rows,), x)
        alpha_r = 1.0 / (self._s1 * self._s1)  # bsi_recon_nll's Normal has std = alpha_R^-1/2
        d = self.discretization
        if d is None:
            N.check(N.lib().bsi_recon_nll(N.ptr(x), N.ptr(x_hat), alpha_r, None, 0.0, 1.0, 0, rows, B, self._D, N.ptr(out), N.stream()))
        else:
            bounds = d.bin_boundaries(x.device, x.dtype).contiguous()
            N.check(N.lib().bsi_recon_nll(N.ptr(x), N.ptr(x_hat), alpha_r, N.ptr(bounds), d.min - d.dx / 2, d.dx, d.k, rows, B,
                                          self._D, N.ptr(out), N.stream()))
        return out.reshape(n_samples, B)

    def discrete_time_loss(self, x: Tensor, n_samples: int, generator=None, *, t: Tensor | None = None) -> Tensor:
        """bfn.py:155-181 -> [n_samples, B]."""
        self._require()
        if t is None:
            # the reference calls the non-existent `self.linspace` here and fails the same way
            raise AttributeError("'BFN' object has no attribute 'linspace' (pass the schedule t explicitly; reference bfn.py:165)")
        t = t.to(torch.float32).contiguous()
        x = x.contiguous()
        n = len(t) - 1
        B = len(x)
        i = torch.randint(0, n, (n_samples, B), device=x.device, generator=generator)
        _, _, wdisc = self._schedule(t)
        t_i = t[i]
        mu = self._sample_flow_distribution(x, t_i, generator)
        x_hat = self._predict_x(mu.flatten(end_dim=1), t_i.flatten(end_dim=1))
        w = wdisc[i].flatten().contiguous()  # sigma_1^((-2/n)(i+1))
        scale = 0.5 * n * (1.0 - self._s1 ** (2.0 / n))
        return _SqErr.apply(x, x_hat, w, scale, False).reshape(n_samples, B)

    def continuous_time_loss(self, x: Tensor, n_samples: int, generator=None) -> Tensor:
        """bfn.py:183-198 -> [n_samples, B]."""
        self._require()
        x = x.contiguous()
        t = self._sample_t(n_samples, len(x), generator)
        mu = self._sample_flow_distribution(x, t, generator)
        tf = t.flatten().contiguous()
        x_hat = self._predict_x(mu.flatten(end_dim=1), tf)
        w = self._coeffs(tf, ("w",))["w"]
        return _SqErr.apply(x, x_hat, w, -math.log(self._s1), False).reshape(t.shape[0], -1)

    def train_loss(self, x: Tensor, generator=None) -> Tensor:
        """bfn.py:200-215: sigma_1^(-2t) * mean_D (x - x_hat)^2, averaged over the batch -> 0-dim."""
        self._require()
        x = x.contiguous()
        t = self._sample_t(1, len(x), generator)[0].contiguous()
        mu = self._sample_flow_distribution(x, t, generator)
        x_hat = self._predict_x(mu, t)
        w = self._coeffs(t, ("w",))["w"]
        return _SqErr.apply(x, x_hat, w, 1.0, True).mean(dim=0)

    # -- sampling (bfn.py:217-280) -------------------------------------------------------------------------
    def _chain(self, n_samples, generator, t, history):
        self._require()
        lib = N.lib()
        if t is None:
            t = torch.linspace(0, 1, self.k + 1, **self.tensor_args)
        t = t.to(torch.float32).contiguous()
        n = t.numel() - 1
        shape = (n_samples, *self.data_shape)
        alpha, rho, _ = self._schedule(t)
        t_eval = torch.cat([t[:n], t.new_ones(1)])  # the final prediction is made at t = 1 (bfn.py:227)
        co = self._coeffs(t_eval, ("c_skip", "c_out"))
        ones = torch.ones(n + 1, **self.tensor_args)
        native = self._native_model()
        mod = native.adaln_table(t_eval) if native is not None else None
        raw = torch.empty(shape, **self.tensor_args)
        x_hat = torch.empty(shape, **self.tensor_args)
        if history:
            mus = torch.empty((n + 1, *shape), **self.tensor_args)
            x_hats = torch.zeros((n + 1, *shape), **self.tensor_args)
            ys = torch.empty((n, *shape), **self.tensor_args)
            mu = mus[0]
            mu.zero_()
        else:
            mu = torch.zeros(shape, **self.tensor_args)
            mu_next = torch.empty(shape, **self.tensor_args)

        def predict(mu_i, i):
            if native is not None:
                native.forward_native(mu_i, mod[i:i + 1], c_in=ones[i:], c_skip=co["c_skip"][i:], c_out=co["c_out"][i:], coef_stride=0,
                                      out=raw)
            else:
                f = self.model(mu_i, t_eval[i].clone().repeat(n_samples)).contiguous()
                N.check(lib.bsi_predict_combine(N.ptr(mu_i), N.ptr(f), N.ptr(co["c_skip"][i:]), N.ptr(co["c_out"][i:]), 0, n_samples,
                                                self._D, N.ptr(raw), N.stream()))
            N.check(lib.bsi_clip(N.ptr(raw), float(self.x_min), float(self.x_max), raw.numel(), N.ptr(x_hat), N.stream()))
            return x_hat

        for i in range(n):
            xh = predict(mu, i)
            eps = torch.randn(shape, **self.tensor_args, generator=generator)
            if history:
                x_hats[i].copy_(xh)
                out_mu, y_o = mus[i + 1], ys[i]
            else:
                out_mu, y_o = mu_next, None
            # y = x_hat + rsqrt(alpha_i) eps;  mu' = (rho_i mu + alpha_i y) / (rho_i + alpha_i): the refine step with lam = rho
            N.check(lib.bsi_refine_step(N.ptr(mu), N.ptr(xh), N.ptr(eps), N.ptr(rho), N.ptr(alpha), None, None, i, 1, n_samples, self._D,
                                        None, N.ptr(y_o), N.ptr(out_mu), N.stream()))
            if history:
                mu = mus[i + 1]
            else:
                mu, mu_next = mu_next, mu
        final = predict(mu, n)
        if history:
            x_hats[n].copy_(final)
            return mus, x_hats, ys
        return final.clone()

    def sample(self, n_samples: int, generator=None, *, t: Tensor | None = None) -> Tensor:
        return self._chain(n_samples, generator, t, history=False)

    def sample_history(self, n_samples: int, generator=None, *, t: Tensor | None = None):
        return self._chain(n_samples, generator, t, history=True)

    # -- pieces (bfn.py:282-329) ---------------------------------------------------------------------------
    def _predict_x(self, mu: Tensor, t: Tensor) -> Tensor:
        self._require()
        mu = mu.contiguous()
        t = t.to(torch.float32).contiguous()
        co = self._coeffs(t, ("c_skip", "c_out"))
        native = self._native_model()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters())
        if native is not None and not needs_grad:
            raw = native.forward_native(mu, native.adaln_table(t), c_in=torch.ones_like(t), c_skip=co["c_skip"], c_out=co["c_out"],
                                        coef_stride=1)
        elif native is not None and hasattr(native, "forward_train"):
            raw = native.forward_train(mu, t, torch.ones_like(t), co["c_skip"], co["c_out"])
        else:
            raw = _PredictCombine.apply(mu, self.model(mu, t), co["c_skip"], co["c_out"])
        return _Clip.apply(raw, float(self.x_min), float(self.x_max))

    def _sample_flow_distribution(self, x: Tensor, t: Tensor, generator=None) -> Tensor:
        self._require()
        x = x.contiguous()
        t = t.to(torch.float32).contiguous()
        co = self._coeffs(t.flatten(), ("fa", "fb"))
        eps = torch.randn((*t.shape, *self.data_shape), **self.tensor_args, generator=generator)
        out = torch.empty_like(eps)
        N.check(N.lib().bsi_affine_noise(N.ptr(x), N.ptr(co["fa"]), N.ptr(co["fb"]), N.ptr(eps), t.numel(), len(x), self._D,
                                         N.ptr(out), N.stream()))
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
