"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of bsi/bfn.py (class BFN) of the reference with explicit
noise, pinned by golden vectors generated from the reference itself (tools/gen_golden_algos.py -> tests/golden/g10_bfn_*.npz).
Every method cites the reference lines it follows."""
import math

import torch

from .bsi_oracle import bcast


class BFNOracle:
    def __init__(self, f, *, data_shape, sigma_1=1e-3, k=50, x_min=-1.0, x_max=1.0, t_min=1e-6, discretization=None,
                 dtype=torch.float32):
        self.f = f
        self.data_shape = tuple(data_shape)
        self.sigma_1 = torch.as_tensor(sigma_1).to(dtype)  # fp32 buffer cast by .to(dtype) (bfn.py:40)
        self.k, self.x_min, self.x_max, self.t_min = k, x_min, x_max, t_min
        self.discretization = discretization
        self.dtype = dtype
        self.D = math.prod(self.data_shape)

    # bfn.py:282-292
    def predict_x(self, mu, t):
        eps_hat = self.f(mu, t)
        gamma = 1 - self.sigma_1 ** (2 * torch.clamp(t, min=self.t_min))
        x_hat = (mu / bcast(gamma, mu) - bcast(torch.sqrt((1 - gamma) / gamma), eps_hat) * eps_hat).clip(self.x_min, self.x_max)
        return torch.where(bcast(t < self.t_min, x_hat), torch.zeros((), dtype=x_hat.dtype), x_hat)

    # bfn.py:294-309; eps of shape t.shape + data_shape
    def flow_sample(self, x, t, eps):
        x = x[(None,) * (t.ndim - 1)]
        gamma = 1 - self.sigma_1 ** (2 * t)
        return torch.addcmul(bcast(gamma, x) * x, bcast(torch.sqrt(gamma * (1 - gamma)), x), eps)

    # bfn.py:311-325 (low-discrepancy branch).  Draw order: rand(()), randperm(n*B)
    def t_grid(self, offset, perm, n, B):
        total = n * B
        return torch.remainder((perm / (1 + total)).reshape(n, B) + offset, 1)

    # bfn.py:125-153.  Draw: randn(n, B, *shape)
    def reconstruction_loss(self, x, eps):
        n, B = eps.shape[0], len(x)
        t = x.new_ones((n, B))
        mu = self.flow_sample(x, t, eps)
        x_hat = self.predict_x(mu.flatten(end_dim=1), t.flatten(end_dim=1)).reshape(n, B, *self.data_shape)
        sigma = self.sigma_1
        d = self.discretization
        if d is None:
            logp = -((x - x_hat) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))
        else:
            bounds = d.bin_boundaries(x.dtype)
            idx = d.bucketize(x)

            def ncdf(v):  # torch.distributions.Normal.cdf
                return 0.5 * (1 + torch.erf((v - x_hat) * (1.0 / sigma) / math.sqrt(2)))

            cl, cr = ncdf(bounds[idx]), ncdf(bounds[idx + 1])
            cl = torch.where(idx == 0, torch.zeros((), dtype=x.dtype), cl)
            cr = torch.where(idx == d.k - 1, torch.ones((), dtype=x.dtype), cr)
            logp = torch.log(torch.clamp(cr - cl, min=1e-20))
        return (-logp).reshape(n, B, -1).sum(dim=2)

    # bfn.py:155-181.  Draw: randint(0, n, (ns, B)), randn(ns, B, *shape)
    def discrete_time_loss(self, x, i, eps, t):
        n = len(t) - 1
        ns, B = i.shape
        t_i = t[i]
        mu = self.flow_sample(x, t_i, eps)
        x_hat = self.predict_x(mu.flatten(end_dim=1), t_i.flatten(end_dim=1)).reshape(ns, B, *self.data_shape)
        err = (x - x_hat).square().reshape(ns, B, -1).sum(dim=2)
        return 0.5 * n * (1 - (self.sigma_1 ** (2 / n))) * ((self.sigma_1 ** ((-2 / n) * (i + 1))) * err)

    # bfn.py:183-198.  Draw: rand(()), randperm(ns*B), randn(ns, B, *shape)
    def continuous_time_loss(self, x, offset, perm, eps):
        ns, B = eps.shape[0], len(x)
        t = self.t_grid(offset, perm, ns, B)
        mu = self.flow_sample(x, t, eps)
        x_hat = self.predict_x(mu.flatten(end_dim=1), t.flatten(end_dim=1)).reshape(ns, B, *self.data_shape)
        err = (x - x_hat).square().reshape(ns, B, -1).sum(dim=2)
        return -torch.log(self.sigma_1) * ((self.sigma_1 ** (-2 * t)) * err)

    # bfn.py:200-215.  Draw: rand(()), randperm(B), randn(B, *shape)
    def train_loss(self, x, offset, perm, eps):
        t = self.t_grid(offset, perm, 1, len(x))[0]
        mu = self.flow_sample(x, t, eps)
        x_hat = self.predict_x(mu, t)
        err = (x - x_hat).square().reshape(len(x), -1).mean(dim=1)
        return ((self.sigma_1 ** (-2 * t)) * err).mean(dim=0)

    # bfn.py:217-280.  Draw: k x randn(n, *shape)
    def sample_history(self, n, eps_steps, t=None, teacher_mus=None):
        if t is None:
            t = torch.linspace(0, 1, self.k + 1, dtype=torch.float32).to(self.dtype)
        k = len(t) - 1
        mu = torch.zeros((n, *self.data_shape), dtype=self.dtype)
        rho = 1.0
        mus, xs, ys = [mu], [], []
        for i in range(k):
            mu_in = mu if teacher_mus is None else teacher_mus[i]
            x_hat = self.predict_x(mu_in, t[i].clone().repeat(n))
            alpha = self.sigma_1 ** (-2 * t[i + 1]) * (1 - self.sigma_1 ** (2 * (t[i + 1] - t[i])))
            y = x_hat + torch.rsqrt(alpha) * eps_steps[i]
            mu = (rho * mu_in + alpha * y) / (rho + alpha)
            rho = rho + alpha
            xs.append(x_hat); ys.append(y); mus.append(mu)
        mu_in = mu if teacher_mus is None else teacher_mus[k]
        xs.append(self.predict_x(mu_in, mu.new_ones((n,))))
        return torch.stack(mus), torch.stack(xs), torch.stack(ys)
