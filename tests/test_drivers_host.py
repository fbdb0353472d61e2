"""CPU tests of the host logic around the hot path (bsi_amd/drivers.py): sampling schedules, the bpd bookkeeping of the ELBO
evaluation loop, batch splitting and Lightning-checkpoint key mapping.  No kernel runs here: BSI is replaced by stand-ins."""
import json
import math

import numpy as np
import torch

from bsi_amd import drivers as D
from oracle.bsi_oracle import LogUniformOracle


class _FakeBSI:
    """lambda_0 / alpha_M / p_lambda as `bsi_amd.BSI` exposes them (0-dim fp32 tensors, LogUniform)."""

    def __init__(self, lambda_0=1e-2, alpha_M=1e6, data_shape=(3, 4, 4)):
        self.lambda_0 = torch.tensor(lambda_0)
        self.alpha_M = torch.tensor(alpha_M)
        self.p_lambda = LogUniformOracle(lambda_0, alpha_M)
        self.data_shape = data_shape
        self.calls = []

    def _fake(self, x, r, m, seed):
        g = torch.Generator().manual_seed(seed + len(self.calls))
        bpd = 3.0 + 0.1 * torch.randn(len(x), generator=g)
        var = 0.01 * torch.rand(len(x), generator=g)
        return -bpd, bpd, {"bpd_var": var, "l_recon": bpd, "l_measure": bpd}

    def elbo(self, x, r, m, generator=None, *, estimate_var=False):
        assert estimate_var
        self.calls.append(("inf", len(x)))
        return self._fake(x, r, m, 1)

    def finite_elbo(self, x, r, m, generator=None, *, t=None, estimate_var=False):
        assert estimate_var and t is not None
        self.calls.append((len(t) - 1, len(x)))
        assert torch.equal(t, torch.linspace(0.0, 1.0, len(t)))
        return self._fake(x, r, m, 100)

    def sample(self, n, generator=None, t=None):
        return torch.linspace(-1.2, 1.2, n * math.prod(self.data_shape)).reshape(n, *self.data_shape)


def test_sampling_schedules():
    b = _FakeBSI()
    k = 64
    lam0, lam1 = 1e-2, float(torch.tensor(1e-2) + torch.tensor(1e6))
    for name in D.SCHEDULES:
        t = D.sampling_schedule(b, name, k)
        assert t.shape == (k + 1,)
        assert abs(float(t[0])) < 1e-6 and abs(float(t[-1]) - 1) < 1e-6, (name, t[0], t[-1])
        assert bool((t[1:] > t[:-1]).all()), name
    assert torch.equal(D.sampling_schedule(b, "linear", k), torch.linspace(0, 1, k + 1))
    # independent float64 restatement of generate_samples.py:127-149: t = (ln(1/variance) - ln lambda_0) / ln(lambda_M/lambda_0)
    u = np.linspace(0, 1, k + 1)
    vmax, vmin = 1 / lam0, 1 / lam1
    want = {
        "cosine": (vmax - vmin) * np.cos(u * np.pi / 2) ** 2 + vmin,
        "edm": np.linspace(math.sqrt(vmax), math.sqrt(vmin), k + 1) ** 2,
        "edm7": ((vmax ** (1 / 14) + u * (vmin ** (1 / 14) - vmax ** (1 / 14))) ** 7) ** 2,
    }
    for name, var in want.items():
        ref = (np.log(1 / var) - math.log(lam0)) / (math.log(lam1) - math.log(lam0))
        got = D.sampling_schedule(b, name, k).double().numpy()
        assert np.abs(got - ref).max() < 5e-5, (name, np.abs(got - ref).max())  # fp32 variance arithmetic near t = 1
    try:
        D.sampling_schedule(b, "bogus", 4)
        raise AssertionError("unknown schedule accepted")
    except ValueError:
        pass


def test_batch_sizes_and_rank_shares():
    assert D.get_batch_sizes(10, 4) == [4, 4, 2] and D.get_batch_sizes(8, 4) == [4, 4] and D.get_batch_sizes(3, 4) == [3]
    assert [D.rank_share(50000, 8, r) for r in range(8)] == [6250] * 8
    assert [D.rank_share(10, 4, r) for r in range(4)] == [3, 3, 2, 2]
    from bsi_amd import Discretization
    b = _FakeBSI()
    out = D.generate_samples(b, Discretization.image_8bit(), 10, 4, rank=3, world_size=4)
    assert out["samples"].shape == (2, 3, 4, 4) and out["images"].dtype == torch.uint8
    u = Discretization.image_8bit().to_unit_interval(b.sample(2))
    assert torch.equal(out["images"], (255 * u.clamp(0, 1)).to(torch.uint8))
    assert int(out["images"].min()) == 0 and int(out["images"].max()) == 255  # clamped outside [-1, 1]


def test_elbo_loop_bookkeeping_and_results_layout(tmp_path):
    b = _FakeBSI()
    batches = [torch.zeros(5, 3, 4, 4), (torch.zeros(3, 3, 4, 4), torch.zeros(3))]  # bare tensors and (x, label) pairs
    ks = ["inf", 8]
    acc = D.evaluate_elbo(b, batches, 2, 3, ks)
    assert b.calls == [("inf", 5), ("inf", 3), (8, 5), (8, 3)]
    # eval_elbo.py:150-160 restated with numpy on the same per-sample values
    b2 = _FakeBSI()
    for steps, seed in (("inf", 1), (8, 100)):
        means, mvars = np.zeros((0,)), np.zeros((0,))
        for x in (torch.zeros(5), torch.zeros(3)):
            b2.calls.append(None)
            _, bpd, extra = b2._fake(x, 2, 3, seed)
            means = np.concatenate((means, bpd.numpy()))
            mvars = np.concatenate((mvars, extra["bpd_var"].numpy()))
        n = len(means)
        assert abs(acc[steps].mean() - means.mean()) < 1e-12
        assert abs(acc[steps].mean_var() - (means.var(ddof=1) + mvars.mean()) / n) < 1e-15
        assert abs(acc[steps].mc_std() - math.sqrt((means.var(ddof=1) + mvars.mean()) / n)) < 1e-12
    res = D.elbo_results(acc, ckpt="x.ckpt", split="test", r_samples=2, m_samples=3, ks=ks, overrides=["a=b"])
    assert set(res) == {"ckpt", "config", "bpd_means", "bpd_mean_vars"}
    assert res["config"] == {"split": "test", "r_samples": 2, "m_samples": 3, "k": ks, "overrides": ["a=b"]}
    D.write_results(tmp_path / "sub" / "r.json", res)
    back = json.loads((tmp_path / "sub" / "r.json").read_text())
    assert set(back["bpd_means"]) == {"inf", "8"} and back["bpd_means"]["inf"] == res["bpd_means"]["inf"]
    # merging shards equals one pass over all samples
    a, c = D.BpdAccumulator(), D.BpdAccumulator()
    a.add([1.0, 2.0], [0.1, 0.2]); c.add([3.0], [0.3])
    a.merge(c)
    assert a.mean() == 2.0 and abs(a.mean_var() - (1.0 + 0.2) / 3) < 1e-15


def test_lightning_checkpoint_keys_round_trip():
    torch.manual_seed(0)
    m, e = torch.nn.Sequential(torch.nn.Linear(3, 2)), torch.nn.Sequential(torch.nn.Linear(3, 2))
    sd = D.to_lightning_state_dict(m, e, ema_step=1234)
    assert set(sd) == {"model.0.weight", "model.0.bias", "ema_model.ema_model.0.weight", "ema_model.ema_model.0.bias",
                       "ema_model._extra_state"}
    assert sd["ema_model._extra_state"] == {"initted": True, "step": 1234}
    m2, e2 = torch.nn.Sequential(torch.nn.Linear(3, 2)), torch.nn.Sequential(torch.nn.Linear(3, 2))
    extra = D.load_lightning_checkpoint({"state_dict": sd, "config": {}}, m2, e2)
    assert extra == {"initted": True, "step": 1234}
    assert torch.equal(m2[0].weight, m[0].weight) and torch.equal(e2[0].bias, e[0].bias)
    online, ema, _ = D.split_lightning_state_dict({"model.a": 1, "other": 2})
    assert online == {"a": 1} and ema is None
    try:
        D.load_lightning_checkpoint({"state_dict": {"model.0.weight": m[0].weight, "model.0.bias": m[0].bias}}, m2, e2)
        raise AssertionError("missing EMA weights accepted")
    except KeyError:
        pass


# ----------------------------------------------------------------------------------------------------------------------
# Reference-derived fixtures (tests/golden/g12_drivers.npz, tools/gen_golden_drivers.py: the reference's own statements
# executed in the build container)
# ----------------------------------------------------------------------------------------------------------------------
def _g12():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_drivers.npz"))
    return z


def test_schedules_vs_reference_generate_samples():
    """scripts/generate_samples.py:117-152 executed by the generator for linear / cosine / edm / edm7 at k in {8, 128}."""
    z = _g12()
    b = _FakeBSI()
    for k in (8, 128):
        for name in D.SCHEDULES:
            ref = z[f"sched_{name}_{k}"]
            got = D.sampling_schedule(b, name, k).numpy()
            assert got.dtype == ref.dtype == np.float32 and got.shape == ref.shape
            # same fp32 operation sequence; LogUniformOracle.cdf follows bsi.py:80-81
            assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= 2e-7, (name, k)


class _Recorded:
    def __init__(self, z, sizes):
        self.z, self.pos, self.sizes, self.calls = z, {}, sizes, []

    def _next(self, key, x):
        i = self.pos.get(key, 0)
        self.pos[key] = i + 1
        bpd, var = torch.from_numpy(self.z[f"elbo_rec_{key}_{i}_bpd"]), torch.from_numpy(self.z[f"elbo_rec_{key}_{i}_var"])
        assert len(bpd) == len(x)
        return -bpd, bpd, {"bpd_var": var}

    lambda_0 = torch.tensor(1e-2)

    def elbo(self, x, r, m, generator=None, *, estimate_var=False):
        self.calls.append(["inf", len(x), r, m, bool(estimate_var)])
        return self._next("inf", x)

    def finite_elbo(self, x, r, m, generator=None, *, t=None, estimate_var=False):
        self.calls.append([len(t) - 1, len(x), r, m, bool(estimate_var), [float(t[0]), float(t[-1])]])
        return self._next(len(t) - 1, x)


def test_elbo_bookkeeping_vs_reference_eval_elbo():
    """scripts/eval_elbo.py:119-173 (loop + results dictionary) executed by the generator on recorded per-sample arrays:
    same call sequence, same means, same variance of the mean, same JSON."""
    z = _g12()
    sizes = [int(n) for n in z["elbo_batch_sizes"]]
    ks = ["inf", 8, 32]
    fake = _Recorded(z, sizes)
    batches = [(torch.zeros(n, 3, 4, 4), torch.zeros(n)) for n in sizes]
    acc = D.evaluate_elbo(fake, batches, 2, 3, ks)
    assert fake.calls == json.loads(str(z["elbo_calls_json"]))
    for k in ks:
        assert abs(acc[k].mean() - float(z[f"elbo_mean_{k}"])) <= 1e-15 * abs(float(z[f"elbo_mean_{k}"])) + 1e-15
        assert abs(acc[k].mean_var() - float(z[f"elbo_mean_var_{k}"])) <= 1e-12 * float(z[f"elbo_mean_var_{k}"])
    res = D.elbo_results(acc, ckpt="run/last.ckpt", split="test", r_samples=2, m_samples=3, ks=ks, overrides=["a=b"])
    ref = json.loads(str(z["elbo_results_json"]))
    got = json.loads(json.dumps(res))
    assert got["ckpt"] == ref["ckpt"] and got["config"] == ref["config"]
    assert set(got) == set(ref) and set(got["bpd_means"]) == set(ref["bpd_means"]) == {"inf", "8", "32"}
    for sect in ("bpd_means", "bpd_mean_vars"):
        for k in ref[sect]:
            assert abs(got[sect][k] - ref[sect][k]) <= 1e-12 * abs(ref[sect][k])


def test_checkpoint_keys_vs_reference_ema_state_dict():
    """Key list, shapes and `_extra_state` of a reference task-shaped module {model, ema_model = create_ema(model)}
    (bsi/tasks/bsi.py:73-81, ema_pytorch.py:196-201) against `to_lightning_state_dict` / `load_lightning_checkpoint`."""
    z = _g12()
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    mk = lambda: DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8))  # noqa: E731
    torch.manual_seed(1)
    m, e = mk(), mk()
    sd = D.to_lightning_state_dict(m, e, ema_step=5)
    ref_keys = json.loads(str(z["ckpt_keys_json"]))
    assert list(sd.keys()) == ref_keys                       # same keys in the same order
    shapes = json.loads(str(z["ckpt_shapes_json"]))
    assert {k: list(v.shape) for k, v in sd.items() if hasattr(v, "shape")} == shapes
    assert sd["ema_model._extra_state"] == json.loads(str(z["ckpt_extra_state_5_json"]))
    assert D.to_lightning_state_dict(m, e, ema_step=0)["ema_model._extra_state"] == json.loads(str(z["ckpt_extra_state_0_json"]))
    m2, e2 = mk(), mk()
    extra = D.load_lightning_checkpoint({"state_dict": sd}, m2, e2)
    assert extra == {"initted": True, "step": 5}
    for (n1, p1), (_, p2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(p1, p2), n1
