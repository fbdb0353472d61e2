"""Callers either side of the hot path: the ELBO / bits-per-dimension evaluation loop, the
sampling schedules and sample export, and Lightning-checkpoint interop.  Host logic only: every number comes from
`bsi_amd.BSI` (HIP kernels); nothing here computes on tensors beyond bookkeeping.

Reference behaviour restated:
  * scripts/eval_elbo.py:108-173 - per k ("inf" or an int): loop over batches, `elbo(..., estimate_var=True)` or
    `finite_elbo(..., t=linspace(0, 1, k+1))`, running `bpd_means` / `bpd_mean_vars`, final mean and
    variance-of-the-mean `(var(bpd, ddof=1) + mean(bpd_var)) / n`, JSON layout of the results file;
  * scripts/generate_samples.py:117-152,162-171 - "linear" / "cosine" / "edm" / "edm7" schedules as
    `t = p_lambda.cdf(1 / variance)`, batches of `eval_batch_size`, `to_unit_interval`, clamp, `(255 x).to(uint8)`;
  * bsi/lightning/callbacks.py:15-16 + SURVEY Appendix C - checkpoint `state_dict` keys `model.*`,
    `ema_model.ema_model.*`, `ema_model._extra_state`.
"""
import json
import math
from pathlib import Path

import numpy as np
import torch

SCHEDULES = ("linear", "cosine", "edm", "edm7")


def sampling_schedule(bsi, name: str, k: int) -> torch.Tensor:
    """k+1 schedule times in [0, 1] (scripts/generate_samples.py:117-149; `eval_fid.py`'s variants are stale)."""
    dev = bsi.lambda_0.device
    if name == "linear":
        return torch.linspace(0, 1, k + 1, device=dev)
    max_variance = 1 / bsi.lambda_0
    min_variance = 1 / (bsi.lambda_0 + bsi.alpha_M)
    if name == "cosine":
        variance = (max_variance - min_variance) * torch.cos(torch.linspace(0, 1, k + 1, device=dev) * torch.pi / 2) ** 2 \
            + min_variance
    elif name == "edm":
        variance = torch.linspace(float(max_variance.sqrt()), float(min_variance.sqrt()), k + 1, device=dev) ** 2
    elif name == "edm7":
        t = torch.linspace(0, 1, k + 1, device=dev)
        max_std, min_std, rho = max_variance.sqrt(), min_variance.sqrt(), 7
        variance = ((max_std ** (1 / rho) + t * (min_std ** (1 / rho) - max_std ** (1 / rho))) ** rho) ** 2
    else:
        raise ValueError(f"Unknown schedule {name}")
    return bsi.p_lambda.cdf(1 / variance)


def get_batch_sizes(num_samples: int, batch_size: int):
    """Full batches plus one remainder batch (scripts/generate_samples.py:35-41)."""
    sizes = [batch_size] * (num_samples // batch_size)
    if num_samples % batch_size:
        sizes.append(num_samples % batch_size)
    return sizes


def rank_share(num_samples: int, world_size: int, rank: int) -> int:
    """Samples a rank generates when `num_samples` independent chains are split over ranks (no collective)."""
    return num_samples // world_size + int(rank < num_samples % world_size)


def generate_samples(bsi, discretization, num_samples: int, batch_size: int, generator=None, *, t=None, rank: int = 0,
                     world_size: int = 1):
    """This rank's share of `num_samples` samples: {"samples": float32 in the unit interval [n, *shape] (CPU),
    "images": uint8 [n, *shape] (CPU)} as scripts/generate_samples.py:162-171 / eval_fid.py:164-167 produce them."""
    samples, images = [], []
    with torch.inference_mode():
        for bs in get_batch_sizes(rank_share(num_samples, world_size, rank), batch_size):
            batch = discretization.to_unit_interval(bsi.sample(bs, generator=generator, t=t))
            samples.append(batch.cpu())
            images.append((255 * batch.clamp(min=0.0, max=1.0)).to(torch.uint8).cpu())
    shape = tuple(bsi.data_shape)
    return {"samples": torch.cat(samples) if samples else torch.empty((0, *shape)),
            "images": torch.cat(images) if images else torch.empty((0, *shape), dtype=torch.uint8)}


class BpdAccumulator:
    """Running per-sample bpd and per-sample Monte-Carlo variance of one k (scripts/eval_elbo.py:113-160)."""

    def __init__(self):
        self.bpd = np.zeros((0,))
        self.bpd_var = np.zeros((0,))

    def add(self, bpd, bpd_var):
        self.bpd = np.concatenate((self.bpd, np.asarray(bpd, dtype=np.float64).reshape(-1)))
        self.bpd_var = np.concatenate((self.bpd_var, np.asarray(bpd_var, dtype=np.float64).reshape(-1)))

    def merge(self, other: "BpdAccumulator"):
        self.add(other.bpd, other.bpd_var)

    def mean(self) -> float:
        return float(self.bpd.mean())

    def mean_var(self) -> float:
        """Variance of the mean: sample variance across images plus the mean within-image MC variance, over n."""
        n = len(self.bpd)
        return float((self.bpd.var(ddof=1) + self.bpd_var.mean()) / n)

    def mc_std(self) -> float:
        return math.sqrt(self.mean_var())


def evaluate_elbo(bsi, batches, r_samples: int, m_samples: int, ks, generator=None, *, progress=None):
    """bits per dimension for every k in `ks` ("inf" -> `elbo`, int -> `finite_elbo` with the linear schedule) over the
    images yielded by `batches` (an iterable that can be iterated once per k).  Returns {k: BpdAccumulator}."""
    acc = {}
    dev = bsi.lambda_0.device
    with torch.inference_mode():
        for steps in ks:
            a = acc[steps] = BpdAccumulator()
            for x in batches:
                x = x[0] if isinstance(x, (tuple, list)) else x
                x = x.to(dev)
                if steps == "inf":
                    _, bpd, extra = bsi.elbo(x, r_samples, m_samples, generator, estimate_var=True)
                else:
                    t = torch.linspace(0.0, 1.0, int(steps) + 1, device=dev)
                    _, bpd, extra = bsi.finite_elbo(x, r_samples, m_samples, generator, estimate_var=True, t=t)
                a.add(bpd.cpu().numpy(), extra["bpd_var"].cpu().numpy())
                if progress is not None:
                    progress(steps, a)
    return acc


def gather_accumulators(acc, group=None):
    """Merge the per-rank accumulators (each rank evaluated its own shard of the data) on every rank."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return acc
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, {k: (a.bpd, a.bpd_var) for k, a in acc.items()}, group=group)
    out = {}
    for k in acc:
        out[k] = BpdAccumulator()
        for p in parts:
            out[k].add(*p[k])
    return out


def elbo_results(acc, *, ckpt: str, split: str, r_samples: int, m_samples: int, ks, overrides=()):
    """The results dictionary scripts/eval_elbo.py:161-173 writes."""
    return {
        "ckpt": str(ckpt),
        "config": {"split": split, "r_samples": r_samples, "m_samples": m_samples, "k": list(ks), "overrides": list(overrides)},
        "bpd_means": {k: a.mean() for k, a in acc.items()},
        "bpd_mean_vars": {k: a.mean_var() for k, a in acc.items()},
    }


def write_results(path, results):
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    path.write_text(json.dumps(results))


# ---------------------------------------------------------------------------------------------------
# Lightning checkpoint interop
# ---------------------------------------------------------------------------------------------------
MODEL_PREFIX = "model."
EMA_PREFIX = "ema_model.ema_model."
EMA_EXTRA = "ema_model._extra_state"


def split_lightning_state_dict(state_dict):
    """(online weights, EMA weights or None, EMA extra state or None) from a reference checkpoint's `state_dict`."""
    online = {k[len(MODEL_PREFIX):]: v for k, v in state_dict.items() if k.startswith(MODEL_PREFIX)}
    ema = {k[len(EMA_PREFIX):]: v for k, v in state_dict.items() if k.startswith(EMA_PREFIX)}
    return online, (ema or None), state_dict.get(EMA_EXTRA)


def load_lightning_checkpoint(path_or_ckpt, model, ema_model=None, *, strict: bool = True):
    """Load a checkpoint trained with the reference (`ckpt["state_dict"]`, bsi/lightning/callbacks.py:15-16) into the
    native modules.  Returns the EMA extra state ({"initted", "step"}) if present."""
    ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=False) if not isinstance(path_or_ckpt, dict) else path_or_ckpt
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    online, ema, extra = split_lightning_state_dict(sd)
    model.load_state_dict(online, strict=strict)
    if ema_model is not None:
        if ema is None:
            raise KeyError(f"checkpoint has no '{EMA_PREFIX}*' entries")
        ema_model.load_state_dict(ema, strict=strict)
    return extra


def to_lightning_state_dict(model, ema_model=None, *, ema_step: int | None = None):
    """The `state_dict` a reference task would save for these weights (keys of SURVEY Appendix C)."""
    out = {MODEL_PREFIX + k: v.detach().cpu() for k, v in model.state_dict().items()}
    if ema_model is not None:
        if ema_step is not None:
            # EMA.get_extra_state (ema_pytorch.py:196-197); a module's extra state precedes its children's entries
            out[EMA_EXTRA] = {"initted": ema_step > 0, "step": int(ema_step)}
        out.update({EMA_PREFIX + k: v.detach().cpu() for k, v in ema_model.state_dict().items()})
    return out
