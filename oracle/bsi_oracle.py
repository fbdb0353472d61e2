"""CPU oracle of the BSI algorithm wrapper (TEST INFRASTRUCTURE — see oracle/__init__.py).

Functional restatement of ``bsi/bsi.py`` of the reference.  All noise is an
explicit argument (the reference draws it from a ``torch.Generator``; the draw
ORDER per entry point is restated in the docstrings so seeded parity runs can
reproduce it).  ``f`` is the denoiser: ``f(mu[B,*shape], t[B]) -> [B,*shape]``.

Reference lines cited as ``bsi.py:L``.
"""
import math
from dataclasses import dataclass

import torch


# ----------------------------------------------------------------------------------------
# Discretization  (bsi.py:12-58)
# ----------------------------------------------------------------------------------------
@dataclass
class Disc:
    lo: float
    hi: float
    k: int

    @classmethod
    def image_8bit(cls):  # bsi.py:24-27
        return cls(-1.0, 1.0, 256)

    @property
    def dx(self):  # bsi.py:55-58
        return (self.hi - self.lo) / (self.k - 1)

    @property
    def range(self):  # bsi.py:50-53
        return (self.lo - self.dx / 2, self.hi + self.dx / 2)

    def bin_boundaries(self, dtype):  # bsi.py:29-30
        a, b = self.range
        return torch.linspace(a, b, self.k + 1, dtype=dtype)

    def bucketize(self, x):  # bsi.py:32-35 (truncation toward zero, then clamp)
        return ((x - (self.lo - self.dx / 2)) / self.dx).to(torch.int64).clamp(0, self.k - 1)

    def to_unit_interval(self, x):  # bsi.py:37-39
        return (x - self.lo) / (self.hi - self.lo)

    def to_8bit_image(self, x):  # bsi.py:41-48
        return (self.to_unit_interval(x) * 255).clamp(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------------------
# LogUniform over lambda  (bsi.py:67-84, constructed at bsi.py:135)
# ----------------------------------------------------------------------------------------
class LogUniformOracle:
    """``low``/``high`` are the values of the buffers ``lambda_0`` and ``lambda_0 + alpha_M`` AT
    CONSTRUCTION, i.e. in torch's default dtype fp32 (a later ``BSI.to(float64)`` does not
    rebuild ``p_lambda``); the logs are Python doubles exactly as in the reference."""

    def __init__(self, lambda_0: float, alpha_M: float):
        lam0 = torch.as_tensor(lambda_0, dtype=torch.float32)
        high = lam0 + torch.as_tensor(alpha_M, dtype=torch.float32)
        self.low = float(lam0)
        self.high = float(high)
        self.ln_low = math.log(self.low)
        self.ln_high = math.log(self.high)
        self.delta = self.ln_high - self.ln_low

    def icdf(self, q):  # bsi.py:83-84
        return torch.exp(self.delta * q + self.ln_low)

    def cdf(self, v):  # bsi.py:80-81
        return (torch.log(v) - self.ln_low) / self.delta

    def reciprocal_pdf(self, v):  # bsi.py:76-78
        return v * self.delta


def bcast(v, like):  # bsi.py:61-64
    return v.reshape(*v.shape, *((1,) * (like.ndim - v.ndim)))


class BSIOracle:
    """Restatement of ``class BSI`` (bsi.py:87-445) with explicit noise."""

    def __init__(self, f, *, data_shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=50,
                 preconditioning="edm", discretization=None, dtype=torch.float32):
        self.f = f
        self.data_shape = tuple(data_shape)
        self.dtype = dtype
        # buffers are created in the default dtype (fp32) and cast by `.to(dtype)` (bsi.py:127-129)
        self.lambda_0 = torch.as_tensor(lambda_0, dtype=torch.float32).to(dtype)
        self.alpha_M = torch.as_tensor(alpha_M, dtype=torch.float32).to(dtype)
        self.alpha_R = torch.as_tensor(alpha_R, dtype=torch.float32).to(dtype)
        self.k = k
        self.preconditioning = preconditioning
        self.discretization = discretization
        self.p_lambda = LogUniformOracle(lambda_0, alpha_M)
        self.default_schedule = torch.linspace(0.0, 1.0, k + 1, dtype=torch.float32).to(dtype)
        self.D = math.prod(self.data_shape)

    # -- preconditioning (bsi.py:390-403) ---------------------------------------------
    def edm_coeffs(self, t):
        lam = self.p_lambda.icdf(t)
        alpha = lam - self.lambda_0
        kappa = 1 + alpha * (alpha / lam)
        c_skip = alpha / kappa
        c_out = torch.rsqrt(kappa)
        c_in = torch.sqrt(lam / kappa)
        return c_skip, c_out, c_in

    # -- x_hat (bsi.py:375-388) -------------------------------------------------------
    def predict_x(self, mu, t):
        if self.preconditioning is None:
            return self.f(mu, t)
        if self.preconditioning == "edm":
            c_skip, c_out, c_in = self.edm_coeffs(t)
            return torch.addcmul(bcast(c_skip, mu) * mu, bcast(c_out, mu),
                                 self.f(bcast(c_in, mu) * mu, t))
        raise RuntimeError(f"Unknown preconditioning {self.preconditioning}")

    # -- forward process sample (bsi.py:405-420); eps ~ N(0,1) of shape lam.shape+data_shape
    def q_mu_lambda(self, x, lam, eps):
        x = x[(None,) * (lam.ndim - 1)]
        return torch.addcmul(bcast((lam - self.lambda_0) / lam, x) * x,
                             bcast(torch.rsqrt(lam), x), eps)

    # -- low-discrepancy lambda grid (bsi.py:422-440).  Draw order: rand(()) then randperm(n*B)
    def lambda_grid(self, offset, perm, n, B):
        total = n * B
        grid = perm / (1 + total)
        t = torch.remainder(grid.reshape(n, B) + offset, 1)
        return self.p_lambda.icdf(t)

    # -- plain sampling (low_discrepancy_sampling=False, bsi.py:441-445).  Draw: rand((B, n)) -- and THAT transposed shape is
    #    what the reference returns; `u` is the recorded draw
    def lambda_plain(self, u):
        return self.p_lambda.icdf(u)

    # -- train loss on that branch (bsi.py:291-310 with 441-445): `_sample_lambda(1, B)[0]` is row 0 of a (B, 1) grid = ONE lambda of
    #    shape (1,), which broadcasts over the batch together with ONE noise image (draws: rand((B, 1)), randn((1, *shape)))
    def train_loss_plain(self, x, u, eps):
        B = len(x)
        lam = self.lambda_plain(u)[0]
        mu = self.q_mu_lambda(x, lam, eps)
        x_hat = self.predict_x(mu, self.p_lambda.cdf(lam))
        err = (x - x_hat).square().reshape(B, -1).mean(dim=1)
        return self.p_lambda.reciprocal_pdf(lam) * err

    # -- infinite-step measurement loss on that branch (bsi.py:276-289): defined where the (B, n) grid broadcasts against the batch
    #    (n == B); draws: rand((B, n)), randn((B, n, *shape))
    def inf_measurement_loss_plain(self, x, u, eps):
        lam = self.lambda_plain(u)
        n = lam.shape[1]
        mu = self.q_mu_lambda(x, lam, eps)
        t = self.p_lambda.cdf(lam).flatten()
        x_hat = self.predict_x(mu.flatten(end_dim=1), t).reshape(n, -1, *self.data_shape)
        err = (x - x_hat).square().reshape(n, x_hat.shape[1], -1).sum(dim=2)
        return 0.5 * self.p_lambda.reciprocal_pdf(lam) * err

    # -- train loss (bsi.py:291-310).  Draw order: rand(()), randperm(B), randn(B,*shape)
    def train_loss(self, x, offset, perm, eps):
        B = len(x)
        lam = self.lambda_grid(offset, perm, 1, B)[0]
        mu = self.q_mu_lambda(x, lam, eps)
        x_hat = self.predict_x(mu, self.p_lambda.cdf(lam))
        err = (x - x_hat).square().reshape(B, -1).mean(dim=1)
        return self.p_lambda.reciprocal_pdf(lam) * err

    # -- sampler (bsi.py:312-373).  Draw order: randn(n,*shape) for mu0, then k x randn(n,*shape)
    def sample_history(self, eps0, eps_steps, t=None, teacher_mus=None):
        """Returns (mus[k+1], x_hats[k+1], ys[k]).  With ``teacher_mus`` ([k+1,n,...]) every
        step starts from the given mu_i instead of its own (teacher forcing)."""
        if t is None:
            t = self.default_schedule
        lam = self.p_lambda.icdf(t)
        alpha = lam.diff()
        k = len(alpha)
        n = eps0.shape[0]
        mu = torch.rsqrt(lam[0]) * eps0
        mus, x_hats, ys = [mu], [], []
        for i in range(k):
            mu_in = mu if teacher_mus is None else teacher_mus[i]
            x_hat = self.predict_x(mu_in, t[i].clone().repeat(n))
            y = x_hat + torch.rsqrt(alpha[i]) * eps_steps[i]
            mu = (alpha[i] * y + lam[i] * mu_in) / lam[i + 1]
            x_hats.append(x_hat)
            ys.append(y)
            mus.append(mu)
        mu_in = mu if teacher_mus is None else teacher_mus[k]
        x_hats.append(self.predict_x(mu_in, mu.new_ones(n)))
        return torch.stack(mus), torch.stack(x_hats), torch.stack(ys)

    def sample(self, eps0, eps_steps, t=None):
        return self.sample_history(eps0, eps_steps, t)[1][-1]

    # -- reconstruction loss (bsi.py:217-247).  Draw: randn(n,B,*shape)
    def reconstruction_loss(self, x, eps):
        n, B = eps.shape[0], len(x)
        lam_M = x.new_full((n, B), float(self.lambda_0 + self.alpha_M))
        mu = self.q_mu_lambda(x, lam_M, eps).flatten(end_dim=1)
        x_hat = self.predict_x(mu, x.new_ones(n * B)).reshape(n, B, *self.data_shape)
        sigma = torch.rsqrt(self.alpha_R)
        if self.discretization is None:
            # Normal(x_hat, sigma).log_prob(x)
            logp = -((x - x_hat) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))
        else:
            d = self.discretization
            bounds = d.bin_boundaries(x.dtype)
            idx = d.bucketize(x)

            def ncdf(v):  # torch.distributions.Normal.cdf
                return 0.5 * (1 + torch.erf((v - x_hat) * (1.0 / sigma) / math.sqrt(2)))

            cl = ncdf(bounds[idx])
            cr = ncdf(bounds[idx + 1])
            cl = torch.where(idx == 0, torch.zeros((), dtype=x.dtype), cl)
            cr = torch.where(idx == d.k - 1, torch.ones((), dtype=x.dtype), cr)
            logp = torch.log(torch.clamp(cr - cl, min=1e-20))
        return (-logp).reshape(n, B, -1).sum(dim=2)

    # -- infinite-step measurement loss (bsi.py:276-289).  Draw: rand(()), randperm(n*B), randn(n,B,*shape)
    def inf_measurement_loss(self, x, offset, perm, eps):
        n, B = eps.shape[0], len(x)
        lam = self.lambda_grid(offset, perm, n, B)
        mu = self.q_mu_lambda(x, lam, eps)
        t = self.p_lambda.cdf(lam).flatten()
        x_hat = self.predict_x(mu.flatten(end_dim=1), t).reshape(n, B, *self.data_shape)
        err = (x - x_hat).square().reshape(n, B, -1).sum(dim=2)
        return 0.5 * self.p_lambda.reciprocal_pdf(lam) * err

    # -- finite-step measurement loss (bsi.py:249-274).  Draw: randint(0,k,(n,B)), randn(n,B,*shape)
    def finite_measurement_loss(self, x, idx, eps, t=None):
        if t is None:
            t = self.default_schedule
        lam = self.p_lambda.icdf(t)
        alpha = lam.diff()
        k = len(alpha)
        n, B = idx.shape
        mu = self.q_mu_lambda(x, lam[idx], eps)
        x_hat = self.predict_x(mu.flatten(end_dim=1), t[idx].flatten()).reshape(n, B, *self.data_shape)
        err = (x - x_hat).square().reshape(n, B, -1).sum(dim=2)
        return (0.5 * k) * alpha[idx] * err

    # -- ELBO assembly (bsi.py:152-215)
    def assemble_elbo(self, l_recon, l_measure, estimate_var=False):
        elbo = -(l_recon.mean(dim=0) + l_measure.mean(dim=0))
        factor = -1 / (math.log(2) * self.D)
        bpd = factor * elbo
        extra = {"l_recon": l_recon, "l_measure": l_measure}
        if estimate_var:
            nr, nm = l_recon.shape[0], l_measure.shape[0]
            assert nr > 1 and nm > 1, "Need at least two samples of each to estimate variance"
            extra["bpd_var"] = factor ** 2 * (l_recon.var(dim=0, unbiased=True) / nr
                                              + l_measure.var(dim=0, unbiased=True) / nm)
        return elbo, bpd, extra


# ----------------------------------------------------------------------------------------
# Train-step tail: global-norm clip + AdamW + EMA  (config/train.yaml:40,
# config/task/optimizer/adamw.yaml, bsi/tasks/ema_pytorch.py:308-434)
# ----------------------------------------------------------------------------------------
def ema_decay(step, beta=0.9999, update_after_step=1000, inv_gamma=1.0, power=2.0 / 3.0, min_value=0.0):
    """ema_pytorch.py:308-314 — `step` is the EMA's own counter BEFORE the update is applied."""
    epoch = max(step - update_after_step - 1, 0.0)
    if epoch <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** -power
    return min(max(value, min_value), beta)


def clip_adamw_step(params, grads, m, v, step, *, lr, beta1, beta2, eps, weight_decay, max_norm):
    """One step of clip_grad_norm_(max_norm) + torch.optim.AdamW (decoupled decay), in place.
    `step` is the 1-based step count.  Returns the pre-clip global gradient norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0) if max_norm is not None else 1.0
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, m_, v_ in zip(params, grads, m, v):
        g = g * coef
        p.mul_(1 - lr * weight_decay)
        m_.mul_(beta1).add_(g, alpha=1 - beta1)
        v_.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m_, denom, value=-lr / bc1)
    return total
