"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of bsi/vdm.py (class VDM) of the reference with explicit
noise, pinned by golden vectors generated from the reference itself (tools/gen_golden_algos.py -> tests/golden/g9_vdm_*.npz).
Every method cites the reference lines it follows."""
import math

import torch
import torch.nn.functional as F

from .bsi_oracle import bcast


class VDMOracle:
    def __init__(self, f, *, data_shape, snr_min=6.73794699909e-3, snr_max=597195.613793, k=50, discretization=None,
                 dtype=torch.float32):
        self.f = f
        self.data_shape = tuple(data_shape)
        self.k = k
        self.discretization = discretization
        self.dtype = dtype
        # buffers are created in the default dtype (fp32) and cast by `.to(dtype)` (vdm.py:41-46)
        self.gamma_0 = (-torch.as_tensor(snr_max).log()).to(dtype)
        self.gamma_1 = (-torch.as_tensor(snr_min).log()).to(dtype)
        self.D = math.prod(self.data_shape)

    # vdm.py:138-150
    def gamma(self, t):
        return torch.lerp(self.gamma_0, self.gamma_1, t)

    def sigma2(self, t):
        return torch.sigmoid(self.gamma(t))

    def alpha(self, t):
        return torch.sqrt(torch.sigmoid(-self.gamma(t)))

    def snr(self, t):
        return torch.exp(-self.gamma(t))

    # vdm.py:324-329
    def predict_x(self, z_t, t):
        return (z_t - bcast(torch.sqrt(self.sigma2(t)), z_t) * self.f(z_t, t)) / bcast(self.alpha(t), z_t)

    # vdm.py:331-348; eps of shape t.shape + data_shape
    def zt_given_x(self, x, t, eps):
        x = x[(None,) * (t.ndim - 1)]
        return torch.addcmul(bcast(self.alpha(t), x) * x, bcast(torch.sqrt(self.sigma2(t)), x), eps)

    # vdm.py:350-379
    def zs_given_zt_x(self, s, z_t, t, x, eps):
        g_s, g_t = self.gamma(s), self.gamma(t)
        ratio = -torch.expm1(F.softplus(-g_t) - F.softplus(g_t) - F.softplus(-g_s) + F.softplus(g_s))
        mean = (bcast(torch.exp(0.5 * (F.softplus(g_s) - F.softplus(g_t)) + F.softplus(-g_t) - F.softplus(-g_s)), z_t) * z_t
                + bcast(self.alpha(s) * ratio, x) * x)
        std = torch.sqrt(self.sigma2(s) * ratio)
        return torch.addcmul(mean, bcast(std, eps), eps)

    # vdm.py:381-397 (low-discrepancy branch).  Draw order: rand(()), randperm(n*B)
    def t_grid(self, offset, perm, n, B):
        total = n * B
        return torch.remainder((perm / (1 + total)).reshape(n, B) + offset, 1)

    # vdm.py:127-136
    def prior_loss(self, x):
        var_1 = self.sigma2(x.new_ones((1,)))
        return 0.5 * (var_1 + (1 - var_1) * x.square() - torch.log(var_1) - 1).reshape(len(x), -1).sum(dim=1)

    # vdm.py:152-195.  Draw: randn(n, B, *shape)
    def reconstruction_loss(self, x, eps):
        zero = x.new_zeros((1,))
        alpha_0 = self.alpha(zero)
        std = torch.sqrt(self.sigma2(zero))
        z_0 = torch.addcmul(alpha_0 * x, std, eps)
        x_hat = z_0 / alpha_0
        s = std / alpha_0

        def log_prob(v):  # torch.distributions.Normal(x_hat, s).log_prob
            return -((v - x_hat) ** 2) / (2 * s ** 2) - torch.log(s) - math.log(math.sqrt(2 * math.pi))

        d = self.discretization
        if d is None:
            logp = log_prob(x)
        else:
            bounds = d.bin_boundaries(x.dtype)
            centers = (bounds[1:] + bounds[:-1]) / 2
            lp = log_prob(bcast(centers, x_hat[None]))                 # [k, n, B, ...]
            lp = F.log_softmax(lp, dim=0)
            idx = d.bucketize(x)
            logp = torch.gather(lp, 0, idx[None, None].expand(1, eps.shape[0], *idx.shape))[0]
        return (-logp).reshape(eps.shape[0], len(x), -1).sum(dim=2)

    # vdm.py:206-231.  Draw: randint(0, T, (n, B)), randn(n, B, *shape)
    def finite_diffusion_loss(self, x, i, eps, t=None):
        if t is None:
            t = torch.linspace(1.0, 0.0, self.k + 1, dtype=torch.float32).to(self.dtype)
        T = len(t) - 1
        n = i.shape[0]
        s_i, t_i = t[i + 1], t[i]
        z_t = self.zt_given_x(x, t_i, eps)
        x_hat = self.predict_x(z_t.flatten(end_dim=1), t_i.flatten(end_dim=1)).reshape(n, len(x), *self.data_shape)
        err = (x - x_hat).square().reshape(n, len(x), -1).sum(dim=2)
        return 0.5 * T * (self.snr(s_i) - self.snr(t_i)) * err

    # vdm.py:233-249.  Draw: rand(()), randperm(n*B), randn(n, B, *shape)
    def inf_diffusion_loss(self, x, offset, perm, eps):
        n, B = eps.shape[0], len(x)
        t = self.t_grid(offset, perm, n, B)
        z_t = self.zt_given_x(x, t, eps)
        x_hat = self.predict_x(z_t.flatten(end_dim=1), t.flatten(end_dim=1)).reshape(n, B, *self.data_shape)
        err = (x - x_hat).square().reshape(n, B, -1).sum(dim=2)
        dsnr = -self.snr(t) * (self.gamma_0 - self.gamma_1)
        return 0.5 * dsnr * err

    # vdm.py:251-262
    def train_loss(self, x, offset, perm, eps):
        return self.inf_diffusion_loss(x, offset, perm, eps[None]) / self.D

    # vdm.py:264-322.  Draw: randn(n, *shape), then k x randn(n, *shape)
    def sample_history(self, eps0, eps_steps, t=None, teacher_z=None):
        """x_hats[k+1] (last = z_0 / alpha_0).  With teacher_z ([k, n, ...]) step i starts from the given z_t."""
        ts = torch.linspace(1.0, 0.0, self.k + 1, dtype=torch.float32).to(self.dtype) if t is None else t
        n = eps0.shape[0]
        z = eps0
        xs, zs = [], [z]
        for i, (tt, ss) in enumerate(zip(ts[:-1], ts[1:])):
            z_in = z if teacher_z is None else teacher_z[i]
            tt_, ss_ = tt.clone().repeat(n), ss.clone().repeat(n)
            x_hat = self.predict_x(z_in, tt_)
            z = self.zs_given_zt_x(ss_, z_in, tt_, x_hat, eps_steps[i])
            xs.append(x_hat)
            zs.append(z)
        xs.append(z / self.alpha(z.new_zeros((1,))))
        return torch.stack(xs), torch.stack(zs)
