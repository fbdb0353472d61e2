"""Mirror of bsi/bsi.py of the reference (Discretization, broadcast_right, LogUniform, BSI) on the
native kernels of include/bsi_hip.h.

Python here is glue only: it draws noise from the caller's `torch.Generator` in exactly the reference's
order, allocates outputs and enqueues HIP kernels on the current stream.  Every
arithmetic step of the hot path (schedule, forward process, preconditioning, measure/refine update, losses,
discretised likelihood) is a HIP kernel; there is no CPU/torch compute fallback — CPU tensors raise.
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import Literal

import torch
from torch import Tensor, nn

from . import _native as N


@dataclass
class Discretization:
    """A discretization of [min, max] into k bins centred on min + (max-min)*i/(k-1)  (bsi.py:12-58).

    Pure bin geometry (host logic, device-agnostic torch expressions exactly as in the reference); the
    likelihood integration over bins and the uint8 export run in HIP kernels (`bsi_recon_nll`,
    `bsi_to_uint8`) when the data is on the GPU."""

    min: float
    max: float
    k: int

    @classmethod
    def image_8bit(cls):
        return cls(-1.0, 1.0, 256)

    def bin_boundaries(self, device: torch.device, dtype: torch.dtype):
        return torch.linspace(*self.range, self.k + 1, device=device, dtype=dtype)

    def bucketize(self, x: Tensor) -> Tensor:
        dx = self.dx
        return ((x - (self.min - dx / 2)) / dx).to(torch.int64).clamp(0, self.k - 1)

    def to_unit_interval(self, x: Tensor) -> Tensor:
        return (x - self.min) / (self.max - self.min)

    def to_8bit_image(self, data: Tensor) -> Tensor:
        if data.is_cuda and data.dtype == torch.float32:
            data = data.contiguous()
            out = torch.empty(data.shape, dtype=torch.uint8, device=data.device)
            N.check(N.lib().bsi_to_uint8(N.ptr(data), self.min, self.max, data.numel(), N.ptr(out), N.stream()))
            return out
        uint8 = torch.iinfo(torch.uint8)
        return (self.to_unit_interval(data) * 255).clamp(uint8.min, uint8.max).to(torch.uint8)

    @property
    def range(self) -> tuple[float, float]:
        dx = self.dx
        return (self.min - dx / 2, self.max + dx / 2)

    @property
    def dx(self) -> float:
        return (self.max - self.min) / (self.k - 1)


def broadcast_right(x: Tensor, other: Tensor):
    """Unsqueeze `x` to the right so that it broadcasts against `other`  (bsi.py:61-64)."""
    assert other.ndim >= x.ndim
    return x.reshape(*x.shape, *((1,) * (other.ndim - x.ndim)))


class LogUniform:
    """Log-uniform law of lambda on [low, high] (bsi.py:67-84); logs are Python doubles as in the reference.
    The tensor helpers serve outside callers (schedule builders in scripts); BSI itself evaluates
    icdf/cdf inside its kernels."""

    def __init__(self, low: float, high: float):
        self.low = low
        self.high = high
        self.ln_low = math.log(self.low)
        self.ln_high = math.log(self.high)
        self.diff_ln_high_ln_low = self.ln_high - self.ln_low

    def reciprocal_pdf(self, value: Tensor) -> Tensor:
        return value * self.diff_ln_high_ln_low

    def cdf(self, value: Tensor) -> Tensor:
        return (torch.log(value) - self.ln_low) / self.diff_ln_high_ln_low

    def icdf(self, quantile: Tensor) -> Tensor:
        return torch.exp(self.diff_ln_high_ln_low * quantile + self.ln_low)


def _new(shape, like: Tensor):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


class _PredictCombine(torch.autograd.Function):
    """x_hat = c_skip*mu + c_out*f (bsi.py:382-386) with its gradient w.r.t. f."""

    @staticmethod
    def forward(ctx, mu, f, c_skip, c_out):
        rows, D = mu.shape[0], mu[0].numel()
        f = f.contiguous()
        out = torch.empty_like(mu)
        N.check(N.lib().bsi_predict_combine(N.ptr(mu), N.ptr(f), N.ptr(c_skip), N.ptr(c_out), 1, rows, D, N.ptr(out),
                                            N.stream()))
        ctx.save_for_backward(c_skip, c_out)
        ctx.dims = (rows, D)
        return out

    @staticmethod
    def backward(ctx, g):
        c_skip, c_out = ctx.saved_tensors
        rows, D = ctx.dims
        g = g.contiguous()
        gf = torch.empty_like(g)
        N.check(N.lib().bsi_predict_combine_bwd(N.ptr(g), N.ptr(c_skip), N.ptr(c_out), 1, rows, D, N.ptr(gf), None,
                                                N.stream()))
        return None, gf, None, None


class _SqErr(torch.autograd.Function):
    """out[r] = w[r]*scale*reduce_D (x[r % B] - x_hat[r])^2 with its gradient w.r.t. x_hat."""

    @staticmethod
    def forward(ctx, x, x_hat, w, scale, mean):
        rows, B, D = x_hat.shape[0], x.shape[0], x[0].numel()
        x_hat = x_hat.contiguous()
        out = _new((rows,), x)
        N.check(N.lib().bsi_sqerr_rows(N.ptr(x), N.ptr(x_hat), N.ptr(w), scale, int(mean), rows, B, D, N.ptr(out),
                                       N.stream()))
        ctx.save_for_backward(x, x_hat, w)
        ctx.args = (scale, int(mean), rows, B, D)
        return out

    @staticmethod
    def backward(ctx, g):
        x, x_hat, w = ctx.saved_tensors
        scale, mean, rows, B, D = ctx.args
        g = g.contiguous()
        gx = torch.empty_like(x_hat)
        N.check(N.lib().bsi_sqerr_rows_bwd(N.ptr(x), N.ptr(x_hat), N.ptr(w), N.ptr(g), scale, mean, rows, B, D,
                                           N.ptr(gx), N.stream()))
        return None, gx, None, None, None


# The reference's task wraps train_loss / elbo / sample / sample_history in torch.compile by default (config/task/bsi.yaml:14,
# bsi/tasks/bsi.py:130-134).  Here these entry points are sequences of C-ABI kernel launches on raw device pointers (plus two
# autograd.Functions whose backward is a kernel launch): there is nothing for a tracing compiler to fuse, and tracing ctypes calls
# would only graph-break at every launch.  They are therefore OPAQUE to dynamo: `torch.compile(bsi.sample)` returns a wrapper that
# runs the method eagerly -- same bits, same generator draws, no graphs, no recompiles (tests/test_hip_compile.py).
def _opaque(fn):
    """Method decorator: the body runs outside dynamo.  (Not `torch.compiler.disable` applied to the function itself:
    `torch.compile(bound_method)` unwraps a disabled function to its UNBOUND original and drops `self`.)"""
    import functools

    inner = torch.compiler.disable(fn, recursive=True)

    @functools.wraps(fn)
    def method(self, *args, **kwargs):
        return inner(self, *args, **kwargs)

    return method


class BSI(nn.Module):
    """Bayesian Sample Inference (arXiv 2502.07580) — drop-in for `bsi.bsi.BSI` (bsi.py:87-445).

    Same constructor (keyword-only after `model`), attributes (`model`/`set_model`, `k`, `data_shape`,
    `discretization`, `p_lambda`, 0-dim non-persistent buffers `lambda_0`, `alpha_R`, `alpha_M`,
    `default_schedule`), methods and return types.  `model` is kept in a list so it is neither a
    submodule nor part of the state dict (bsi.py:123-125)."""

    def __init__(self, model: nn.Module, *, data_shape: tuple[int, ...], lambda_0: float, alpha_M: float,
                 alpha_R: float, k: int, preconditioning: Literal["edm"] | None,
                 low_discrepancy_sampling: bool = True, discretization: Discretization | None = None):
        super().__init__()
        self._model = [model]
        self.data_shape = tuple(data_shape)
        self.register_buffer("lambda_0", torch.as_tensor(lambda_0), persistent=False)
        self.register_buffer("alpha_R", torch.as_tensor(alpha_R), persistent=False)
        self.register_buffer("alpha_M", torch.as_tensor(alpha_M), persistent=False)
        self.k = k
        self.preconditioning = preconditioning
        self.low_discrepancy_sampling = low_discrepancy_sampling
        self.discretization = discretization
        self.p_lambda = LogUniform(self.lambda_0, self.lambda_0 + self.alpha_M)
        self.register_buffer("default_schedule", torch.linspace(0.0, 1.0, self.k + 1), persistent=False)
        # host copy of the scalars for the kernels (fp32 values of the buffers at construction, bsi.py:135)
        self._params = N.BSIParams(float(self.lambda_0), float(self.alpha_M), float(self.alpha_R),
                                   self.p_lambda.ln_low, self.p_lambda.diff_ln_high_ln_low)
        self._D = math.prod(self.data_shape)

    # -- reference surface ---------------------------------------------------------------------------
    @property
    def model(self):
        return self._model[0]

    def set_model(self, model):
        self._model[0] = model

    @property
    def tensor_args(self):
        return {"device": self.lambda_0.device, "dtype": self.lambda_0.dtype}

    # -- helpers ---------------------------------------------------------------------------------------
    def _p(self):
        return C.byref(self._params)

    def _require_fp32(self):
        if self.lambda_0.dtype != torch.float32:
            raise RuntimeError("bsi_amd.BSI: the native path computes the wrapper in fp32 (got "
                               f"{self.lambda_0.dtype})")
        if self.lambda_0.device.type != "cuda":
            raise RuntimeError("bsi_amd.BSI: module is not on a HIP device; there is no CPU path")

    def _native_model(self):
        m = self.model
        return m if hasattr(m, "forward_native") and hasattr(m, "adaln_table") else None

    def _schedule(self, t: Tensor):
        """lam = icdf(t), alpha = diff(lam), EDM coefficients over the schedule (bsi.py:322-323,396-403)."""
        t = t.to(torch.float32).contiguous()
        k1 = t.numel()
        lam, alpha = _new((k1,), t), _new((max(k1 - 1, 1),), t)
        N.check(N.lib().bsi_schedule(self._p(), N.ptr(t), k1, N.ptr(lam), N.ptr(alpha), N.stream()))
        return lam, alpha

    def _coeffs(self, t: Tensor):
        t = t.to(torch.float32).contiguous()
        n = t.numel()
        cs, co, ci = _new((n,), t), _new((n,), t), _new((n,), t)
        N.check(N.lib().bsi_edm_coeffs(self._p(), N.ptr(t), n, None, N.ptr(cs), N.ptr(co), N.ptr(ci), N.stream()))
        return cs, co, ci

    # -- ELBO (bsi.py:152-215) -----------------------------------------------------------------------
    def _assemble(self, l_recon, l_measure, n_recon_samples, n_measure_samples, estimate_var):
        elbo = -(l_recon.mean(dim=0) + l_measure.mean(dim=0))
        conversion_factor = -1 / (math.log(2) * math.prod(self.data_shape))
        bpd = conversion_factor * elbo
        extra = {"l_recon": l_recon, "l_measure": l_measure}
        if estimate_var:
            assert n_recon_samples > 1 and n_measure_samples > 1, (
                "Need at least two samples of each to estimate variance")
            l_recon_var = l_recon.var(dim=0, unbiased=True) / n_recon_samples
            l_measure_var = l_measure.var(dim=0, unbiased=True) / n_measure_samples
            extra["bpd_var"] = (conversion_factor**2) * (l_recon_var + l_measure_var)
        return elbo, bpd, extra

    @_opaque
    def elbo(self, x: Tensor, n_recon_samples: int, n_measure_samples: int, generator=None, *,
             estimate_var: bool = False):
        """Monte Carlo estimate of the infinite-step ELBO: returns (elbo[B], bpd[B], extra)."""
        l_recon = self.reconstruction_loss(x, n_recon_samples, generator)
        l_measure = self.inf_measurement_loss(x, n_measure_samples, generator)
        return self._assemble(l_recon, l_measure, n_recon_samples, n_measure_samples, estimate_var)

    @_opaque
    def finite_elbo(self, x: Tensor, n_recon_samples: int, n_measure_samples: int, generator=None, *,
                    t: Tensor | None = None, estimate_var: bool = False):
        """Monte Carlo estimate of the finite-step ELBO."""
        l_recon = self.reconstruction_loss(x, n_recon_samples, generator)
        l_measure = self.finite_measurement_loss(x, n_measure_samples, generator, t=t)
        return self._assemble(l_recon, l_measure, n_recon_samples, n_measure_samples, estimate_var)

    def reconstruction_loss(self, x: Tensor, n_samples: int, generator=None) -> Tensor:
        """bsi.py:217-247 -> [n_samples, B]."""
        self._require_fp32()
        x = x.contiguous()
        B = len(x)
        lam_M = x.new_full((n_samples, B), float(self.lambda_0 + self.alpha_M))
        mu = self._sample_q_mu_lambda(x, lam_M, generator).flatten(end_dim=1)
        x_hat = self._predict_x(mu, x.new_ones(n_samples * B)).contiguous()
        out = _new((n_samples * B,), x)
        d = self.discretization
        if d is None:
            N.check(N.lib().bsi_recon_nll(N.ptr(x), N.ptr(x_hat), float(self.alpha_R), None, 0.0, 1.0, 0,
                                          n_samples * B, B, self._D, N.ptr(out), N.stream()))
        else:
            bounds = d.bin_boundaries(x.device, x.dtype).contiguous()
            N.check(N.lib().bsi_recon_nll(N.ptr(x), N.ptr(x_hat), float(self.alpha_R), N.ptr(bounds),
                                          d.min - d.dx / 2, d.dx, d.k, n_samples * B, B, self._D, N.ptr(out),
                                          N.stream()))
        return out.reshape(n_samples, B)

    def finite_measurement_loss(self, x: Tensor, n_samples: int, generator=None, *, t: Tensor | None = None):
        """bsi.py:249-274 -> [n_samples, B]."""
        self._require_fp32()
        if t is None:
            t = self.default_schedule
        x = x.contiguous()
        lambda_, alpha = self._schedule(t)
        B = len(x)
        k = len(t) - 1
        i = torch.randint(0, k, (n_samples, B), device=x.device, generator=generator)
        mu = self._sample_q_mu_lambda(x, lambda_[i], generator)
        x_hat = self._predict_x(mu.flatten(end_dim=1), t[i].flatten(end_dim=1))
        w = alpha[i].flatten().contiguous()
        out = _SqErr.apply(x, x_hat, w, 0.5 * k, False)
        return out.reshape(n_samples, B)

    def inf_measurement_loss(self, x: Tensor, n_samples: int, generator=None) -> Tensor:
        """bsi.py:276-289 -> [n_samples, B]."""
        self._require_fp32()
        x = x.contiguous()
        B = len(x)
        lambda_ = self._sample_lambda(n_samples, B, generator)
        mu = self._sample_q_mu_lambda(x, lambda_, generator)
        t, rpdf = self._lambda_to_t(lambda_.flatten())
        x_hat = self._predict_x(mu.flatten(end_dim=1), t)
        out = _SqErr.apply(x, x_hat, rpdf, 0.5, False)
        return out.reshape(n_samples, B)

    @_opaque
    def train_loss(self, x: Tensor, generator=None) -> Tensor:
        """bsi.py:291-310: Delta*lambda*mean_D (x - x_hat)^2, one lambda per batch element -> [B]."""
        self._require_fp32()
        x = x.contiguous()
        lambda_ = self._sample_lambda(1, len(x), generator)[0]
        mu = self._sample_q_mu_lambda(x, lambda_, generator)
        t, rpdf = self._lambda_to_t(lambda_)
        x_hat = self._predict_x(mu, t)
        if rpdf.numel() == 1 and len(x) > 1:  # plain sampling (bsi.py:441-445): row 0 of the (batch, 1) grid is ONE lambda for the batch
            rpdf = rpdf.expand(len(x)).contiguous()
        return _SqErr.apply(x, x_hat, rpdf, 1.0, True)

    # -- sampling (bsi.py:312-373) -------------------------------------------------------------------
    @_opaque
    def sample(self, n_samples: int, generator=None, *, t: Tensor | None = None, graph: bool = False,
               device_noise: bool = False) -> Tensor:
        """Draw `n_samples` samples (Algorithm 3): k+1 denoiser evaluations.

        `graph=True` replays the whole chain as ONE captured HIP graph (captured on first use for this (n_samples, schedule)):
        every launch of the library is stream-capture safe.  The noise is drawn with the same generator calls as the eager
        path, so both paths return bit-identical samples for the same generator state.  Measured on MI355X: no gain at any batch
        size (1..16 images: 4.5 ms per denoiser step either way) -- the small-batch step is bound by the latency of its ~320
        dependent kernels themselves (a 256 x 256 GEMM tile walks its K loop in ~10 us however small M is), not by launch
        overhead; large batches keep the GPU busy anyway.

        `device_noise=True` (opt-in): the Gaussian noise of mu_0 and of every measurement is generated inside the HIP kernels
        (Philox4x32-10 + Box-Muller, bsi_refine_step_philox) from ONE 64-bit seed drawn from `generator` -- no eps tensors, no
        k+1 generator calls.  It is a different random stream from the reference's `torch.randn` calls (bsi.py:325,332-334),
        which is why it is not the default; samples are reproducible for a given generator state."""
        if device_noise:
            if graph:
                raise RuntimeError("BSI.sample: device_noise and graph cannot be combined")
            return self._run_chain(n_samples, generator, t, history=False, device_noise=True)
        if graph:
            return self._run_chain_graphed(n_samples, generator, t)
        return self._run_chain(n_samples, generator, t, history=False)

    def _run_chain_graphed(self, n, generator, t):
        if t is None:
            t = self.default_schedule
        dev = self.lambda_0.device
        native = self._native_model() if self.preconditioning == "edm" else None
        if native is None:
            raise RuntimeError("BSI.sample(graph=True) needs a native denoiser with EDM preconditioning")
        k = t.numel() - 1
        shape = (n, *self.data_shape)
        cache = self.__dict__.setdefault("_graph_cache", {})
        # The captured graph bakes in raw device pointers of the bf16 weight shadows (the "pack") and of the engine
        # workspace.  The entry OWNS both (they stay alive as long as the graph does) and the key names the pack object:
        # a raw-kernel weight update (DPTrainer invalidates the pack without touching parameter versions) or a pack rebuilt
        # for any other reason gives a new object -> a new capture, never a replay over freed or stale memory.
        pack = native.native_pack()
        key = (n, k, t.data_ptr(), t._version, native._weights_key(), id(pack))
        entry = cache.get(key)
        if entry is None:
            cache.clear()  # one graph at a time: its private memory pool holds every intermediate of the chain
            eps0 = torch.empty(shape, **self.tensor_args)
            eps_steps = torch.empty((k, *shape), **self.tensor_args)
            eps0.normal_()
            eps_steps.normal_()
            t_static = t.detach().clone()
            shared_ws = native._ws
            native._ws = None  # the chain allocates a workspace of its own, which the entry keeps
            # warm-up on a side stream (lazy initialisation, function attributes, allocator pools), then capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                self._run_chain(n, None, t_static, history=False, noise=(eps0, eps_steps))
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ws = native._ws
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                out = self._run_chain(n, None, t_static, history=False, noise=(eps0, eps_steps))
            assert native._ws is ws and native.native_pack() is pack
            if shared_ws is not None and shared_ws.numel() >= ws.numel():
                native._ws = shared_ws  # eager calls go back to the shared workspace
            entry = cache[key] = (g, eps0, eps_steps, out, (t_static, pack, ws))
        g, eps0, eps_steps, out, _ = entry
        torch.randn(shape, **self.tensor_args, generator=generator, out=eps0)  # the eager path's draws, in the same order
        for i in range(k):
            torch.randn(shape, **self.tensor_args, generator=generator, out=eps_steps[i])
        g.replay()
        return out.clone()

    @_opaque
    def sample_history(self, n_samples: int, generator=None, *, t: Tensor | None = None):
        """As `sample`, returning (mus[k+1], x_hats[k+1], ys[k])."""
        return self._run_chain(n_samples, generator, t, history=True)

    def _run_chain(self, n, generator, t, history, noise=None, device_noise=False):
        self._require_fp32()
        if t is None:
            t = self.default_schedule
        lib = N.lib()
        dev = self.lambda_0.device
        shape = (n, *self.data_shape)
        D = self._D
        t = t.to(torch.float32).contiguous()
        lam, alpha = self._schedule(t)
        k = t.numel() - 1
        # coefficient table over [t_0..t_{k-1}, 1]: the last prediction is made at t = 1 (bsi.py:336)
        t_eval = torch.cat([t[:k], t.new_ones(1)])
        c_skip, c_out, c_in = self._coeffs(t_eval)
        native = self._native_model() if self.preconditioning == "edm" else None
        mod = native.adaln_table(t_eval) if native is not None else None

        seed = None
        if device_noise:
            assert not history and noise is None
            seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev, generator=generator)  # stays on the device
            eps = torch.empty(shape, **self.tensor_args)
            N.check(lib.bsi_philox_normal(N.ptr(seed), 0xFFFFFFFF, eps.numel(), N.ptr(eps), N.stream()))
        else:
            eps = noise[0] if noise is not None else torch.randn(shape, **self.tensor_args, generator=generator)
        if history:
            mus = torch.empty((k + 1, *shape), **self.tensor_args)
            x_hats = torch.zeros((k + 1, *shape), **self.tensor_args)
            ys = torch.empty((k, *shape), **self.tensor_args)
            mu = mus[0]
        else:
            mu = torch.empty(shape, **self.tensor_args)
            mu_next = torch.empty(shape, **self.tensor_args)
        N.check(lib.bsi_sample_init(N.ptr(eps), N.ptr(lam), n, D, N.ptr(mu), N.stream()))
        x_hat = torch.empty(shape, **self.tensor_args)

        def predict(mu_i, i):
            if native is not None:
                native.forward_native(mu_i, mod[i:i + 1], c_in=c_in[i:], c_skip=c_skip[i:], c_out=c_out[i:],
                                      coef_stride=0, out=x_hat)
                return x_hat, 1
            if self.preconditioning is None:
                return self.model(mu_i, t_eval[i].clone().repeat(n)).contiguous(), 1
            if self.preconditioning == "edm":
                inp = torch.empty_like(mu_i)
                N.check(lib.bsi_scale_rows(N.ptr(mu_i), N.ptr(c_in[i:]), 0, n, D, N.ptr(inp), N.stream()))
                return self.model(inp, t_eval[i].clone().repeat(n)).contiguous(), 0
            raise RuntimeError(f"Unknown preconditioning {self.preconditioning}")

        for i in range(k):
            f, is_xhat = predict(mu, i)
            if history:
                out_mu, xh_o, y_o = mus[i + 1], x_hats[i], ys[i]
            else:
                out_mu, xh_o, y_o = mu_next, None, None
            if device_noise:
                N.check(lib.bsi_refine_step_philox(N.ptr(mu), N.ptr(f), N.ptr(seed), N.ptr(lam), N.ptr(alpha), N.ptr(c_skip),
                                                   N.ptr(c_out), i, is_xhat, n, D, N.ptr(xh_o), N.ptr(y_o), N.ptr(out_mu),
                                                   N.stream()))
            else:
                eps = noise[1][i] if noise is not None else torch.randn(shape, **self.tensor_args, generator=generator)
                N.check(lib.bsi_refine_step(N.ptr(mu), N.ptr(f), N.ptr(eps), N.ptr(lam), N.ptr(alpha), N.ptr(c_skip),
                                            N.ptr(c_out), i, is_xhat, n, D, N.ptr(xh_o), N.ptr(y_o), N.ptr(out_mu),
                                            N.stream()))
            if history:
                mu = mus[i + 1]
            else:
                mu, mu_next = mu_next, mu
        f, is_xhat = predict(mu, k)
        if is_xhat:
            final = f
        else:
            final = torch.empty(shape, **self.tensor_args)
            N.check(lib.bsi_predict_combine(N.ptr(mu), N.ptr(f), N.ptr(c_skip[k:]), N.ptr(c_out[k:]), 0, n, D,
                                            N.ptr(final), N.stream()))
        if history:
            x_hats[k].copy_(final)
            return mus, x_hats, ys
        return final

    # -- pieces (bsi.py:375-445) ---------------------------------------------------------------------
    def _predict_x(self, mu: Tensor, t: Tensor) -> Tensor:
        if t.ndim == 1 and t.numel() == 1 and mu.shape[0] > 1:
            # one time for the whole batch: the reference's coefficient / embedding broadcasting (bsi.py:381-386, dit.py:177)
            t = t.expand(mu.shape[0]).contiguous()
        if self.preconditioning is None:
            return self.model(mu, t)
        elif self.preconditioning == "edm":
            self._require_fp32()
            mu = mu.contiguous()
            c_skip, c_out, c_in = self._edm_preconditioning(t)
            native = self._native_model()
            needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters())
            if native is not None and not needs_grad:
                mod = native.adaln_table(t)
                return native.forward_native(mu, mod, c_in=c_in, c_skip=c_skip, c_out=c_out, coef_stride=1)
            if native is not None and hasattr(native, "forward_train"):
                # training: fused preconditioning + tape-recording forward, hand-written backward
                return native.forward_train(mu, t, c_in, c_skip, c_out)
            rows, D = mu.shape[0], self._D
            inp = torch.empty_like(mu)
            N.check(N.lib().bsi_scale_rows(N.ptr(mu), N.ptr(c_in), 1, rows, D, N.ptr(inp), N.stream()))
            return _PredictCombine.apply(mu, self.model(inp, t), c_skip, c_out)
        else:
            raise RuntimeError(f"Unknown preconditioning {self.preconditioning}")

    def _edm_preconditioning(self, t: Tensor | None = None):
        """(c_skip, c_out, c_in) of the EDM-style preconditioning (bsi.py:390-403)."""
        self._require_fp32()
        return self._coeffs(t)

    def _lambda_to_t(self, lam: Tensor):
        lam = lam.contiguous()
        n = lam.numel()
        t, rpdf = _new((n,), lam), _new((n,), lam)
        N.check(N.lib().bsi_lambda_to_t(self._p(), N.ptr(lam), n, N.ptr(t), N.ptr(rpdf), N.stream()))
        return t, rpdf

    def _sample_q_mu_lambda(self, x: Tensor, lambda_: Tensor, generator=None) -> Tensor:
        """mu_lambda = ((lambda - lambda_0)/lambda) x + lambda^-1/2 eps  (bsi.py:405-420)."""
        self._require_fp32()
        x = x.contiguous()
        lambda_ = lambda_.to(torch.float32).contiguous()
        eps = torch.randn((*lambda_.shape, *self.data_shape), **self.tensor_args, generator=generator)
        # The reference broadcasts lambda [..., batch] against x[None, ..., batch] (bsi.py:411-419).  Every caller of the
        # low-discrepancy branch passes [..., batch] itself; the plain branch's transposed grid (bsi.py:441-445) can pass ONE lambda
        # (train_loss) -- then lambda and its ONE noise image are shared by the batch, as there.  Shapes torch cannot broadcast raise.
        lead = torch.broadcast_shapes(tuple(lambda_.shape), (1,) * max(lambda_.ndim - 1, 0) + (len(x),))
        if tuple(lead) != tuple(lambda_.shape):
            eps = eps.expand(*lead, *self.data_shape).contiguous()
            lambda_ = lambda_.expand(lead).contiguous()
        mu = torch.empty_like(eps)
        N.check(N.lib().bsi_q_sample(self._p(), N.ptr(x), N.ptr(lambda_), N.ptr(eps), lambda_.numel(), len(x),
                                     self._D, N.ptr(mu), N.stream()))
        return mu

    def _sample_lambda(self, n_samples: int, batch_size: int, generator=None) -> Tensor:
        """bsi.py:422-445.  Low-discrepancy branch -> [n_samples, batch]; the plain branch keeps the
        reference's (batch, n_samples) shape."""
        self._require_fp32()
        lib = N.lib()
        if self.low_discrepancy_sampling:
            offset = torch.rand((), **self.tensor_args, generator=generator)
            total = n_samples * batch_size
            perm = torch.randperm(total, device=self.tensor_args["device"], generator=generator)
            lam = torch.empty((n_samples, batch_size), **self.tensor_args)
            N.check(lib.bsi_lambda_grid(self._p(), N.ptr(perm), N.ptr(offset), total, N.ptr(lam), N.stream()))
            return lam
        else:
            t = torch.rand((batch_size, n_samples), **self.tensor_args, generator=generator)
            lam = torch.empty_like(t)
            N.check(lib.bsi_edm_coeffs(self._p(), N.ptr(t), t.numel(), N.ptr(lam), None, None, None, N.stream()))
            return lam
