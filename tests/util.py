"""Helpers shared by the tests: golden-fixture loading and error metrics."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RECORDS = []  # parity records of this pytest process (report())


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if not k.startswith("_meta")}


def weights(tag, device=None, dtype=None):
    w = golden("w_" + tag)
    return {k: (v.to(device=device, dtype=dtype) if v.is_floating_point() else v.to(device=device))
            for k, v in w.items()}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_linf(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b, floor=0.0):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b).abs() / b.abs().clamp_min(floor if floor > 0 else 1e-30)).max())


CALIB = dict(shape=(3, 32, 32), patch_size=4, dim=128, depth=4, heads=2, ff=(6, 8), seed=11, B=64)


def calib_weights():
    """Weights of the SURVEY Appendix F calibration model (g11_calib_dit): a seeded torch-CPU stream from
    oracle.dit_oracle.dit_random_weights, checked against the fingerprint the golden generator stored."""
    from oracle import dit_oracle as do
    c = CALIB
    W = do.dit_random_weights(c["shape"], c["patch_size"], c["dim"], c["depth"], ff=c["ff"], seed=c["seed"])
    fp = torch.stack([torch.stack((v.double().sum(), v.double().abs().sum(), v.flatten()[0].double(),
                                   v.flatten()[-1].double())) for _, v in sorted(W.items())])
    ref = golden("g11_calib_dit")["weight_fingerprint"]
    assert torch.allclose(fp, ref, rtol=1e-12, atol=0), "seeded weight stream differs from the one the golden was made with"
    return W


def report(name, **values):
    """Record achieved parity errors: printed (visible with -rA / -s) and appended to gpurun_out/parity_report.jsonl so
    that the numbers behind a green assertion are on file (copied to profiles/ per round)."""
    import json
    rec = {"test": name, **{k: (float(v) if not isinstance(v, (str, int, list, dict)) else v) for k, v in values.items()}}
    line = json.dumps(rec)
    RECORDS.append(rec)  # printed again by conftest.pytest_terminal_summary: the -q tail of a run carries the achieved errors
    print("PARITY " + line)
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return rec


CALIB_UNET = dict(shape=(3, 16, 16), dim=128, levels=2, ff=(6, 8), seed=13, B=64)


def calib_unet_weights():
    """Weights of the UNet calibration model (g13_calib_unet), checked against the stored fingerprint."""
    from oracle import unet_oracle as uo
    c = CALIB_UNET
    W = uo.unet_random_weights(c["shape"], c["dim"], c["levels"], seed=c["seed"], ff=c["ff"])
    fp = torch.stack([torch.stack((v.double().sum(), v.double().abs().sum(), v.flatten()[0].double(),
                                   v.flatten()[-1].double())) for _, v in sorted(W.items())])
    ref = golden("g13_calib_unet")["weight_fingerprint"]
    assert torch.allclose(fp, ref, rtol=1e-12, atol=0), "seeded weight stream differs from the one the golden was made with"
    return W


import contextlib
from unittest import mock


@contextlib.contextmanager
def replay_draws(device, **queues):
    """Feed recorded draws to torch.rand / randn / randperm / randint calls made inside the block, in order, moved to `device`
    (the goldens and the multi-process tests draw from seeded CPU generators; the device generator has another stream)."""
    qs = {k: list(v) for k, v in queues.items()}

    def pop(name):
        def f(*a, **kw):
            return qs[name].pop(0).to(kw.get("device", device))
        return f

    with contextlib.ExitStack() as st:
        for name in qs:
            st.enter_context(mock.patch.object(torch, name, side_effect=pop(name)))
        yield
    assert all(len(v) == 0 for v in qs.values()), "not all recorded draws were consumed"


def shard_draws(rank, step, nb, shape):
    """The (offset, perm, eps) a rank's train_loss consumes at optimizer step `step` in the two-process tests: a seeded CPU
    stream per (rank, step), so that a single process can restate both ranks' work."""
    g = torch.Generator().manual_seed(1000 + 17 * rank + step)
    return torch.rand((), generator=g), torch.randperm(nb, generator=g), torch.randn((nb, *shape), generator=g)


BOUNDS = {}  # tag -> (worst achieved value, limit) of this pytest process


def bound(tag, value, limit):
    """assert value < limit, and remember the worst achieved value per tag: the terminal summary lists achieved / limit for every
    tag, so that an assertion bound far above what the kernels achieve is visible in the log."""
    assert not isinstance(value, (bool, np.bool_)), (tag, "bound() takes the achieved error, not a comparison result")
    value = float(value)
    w = BOUNDS.get(tag, (0.0, limit))[0]
    BOUNDS[tag] = (max(w, value), limit)
    assert value < limit, (tag, value, limit)
