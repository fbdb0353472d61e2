"""GPU: the in-kernel Gaussian noise generator (row n1: north_star's "Gaussian measurement noise", the opt-in replacement of
torch.randn at /root/reference/bsi/bsi.py:325,332-334) against the published Philox4x32-10 known-answer vectors and against
the CPU restatement oracle/philox_oracle.py -- integers bit for bit, normals within the stated absolute tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import philox_oracle as po
from tests.util import report

pytestmark = pytest.mark.gpu
DEV = "cuda"
KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "philox_kat.json")))


@pytest.fixture(scope="module")
def N():
    from bsi_amd import _native
    _native.lib()
    return _native


def test_raw_blocks_match_random123_known_answers(N):
    ck = torch.tensor([[int(w, 16) for w in v["counter"] + v["key"]] for v in KAT["vectors"]], dtype=torch.int64)
    exp = torch.tensor([[int(w, 16) for w in v["expected"]] for v in KAT["vectors"]], dtype=torch.int64)
    d = torch.from_numpy(ck.numpy().astype(np.uint32).view(np.int32)).to(DEV)
    out = torch.empty((len(KAT["vectors"]), 4), dtype=torch.int32, device=DEV)
    N.check(N.lib().bsi_philox4x32_10(N.ptr(d), d.shape[0], N.ptr(out), N.stream()))
    got = out.cpu().numpy().view(np.uint32).astype(np.int64)
    assert np.array_equal(got, exp.numpy()), [[f"{x:08x}" for x in r] for r in got]


@pytest.mark.parametrize("seed,stream", [(0x0123456789ABCDEF, 0), (77, 5), ((1 << 62) - 1, po.MU0_STREAM)])
def test_stream_matches_the_restatement(N, seed, stream):
    n = 1 << 18
    sd = torch.tensor([seed], dtype=torch.int64, device=DEV)
    u = torch.empty(n, dtype=torch.int32, device=DEV)
    N.check(N.lib().bsi_philox_uint32(N.ptr(sd), stream, n, N.ptr(u), N.stream()))
    ref_u = po.stream_uint32(seed, stream, n)
    assert np.array_equal(u.cpu().numpy().view(np.uint32), ref_u)  # the integer stream: bit for bit
    z = torch.empty(n, device=DEV)
    N.check(N.lib().bsi_philox_normal(N.ptr(sd), stream, n, N.ptr(z), N.stream()))
    ref = po.normals_from_uint32(ref_u)
    err = np.abs(z.cpu().numpy().astype(np.float64) - ref)
    # fp32 Box-Muller on the hardware's log2 / sin / cos units against float64: |z| <= 5.77, absolute error bound 4e-6
    # (one fp32 ulp at |z| in [4, 8) is 4.8e-7; the sine / cosine units carry ~1e-6 absolute error)
    report("philox_normal_vs_fp64", seed=seed, stream=stream, max_abs=float(err.max()), mean_abs=float(err.mean()))
    assert err.max() < 4e-6


def test_device_noise_chain_uses_these_streams(N):
    """BSI.sample(device_noise=True): mu_0 comes from stream 0xFFFFFFFF and step i from stream i of ONE seed drawn from the
    caller's generator -- the chain equals the default chain fed with the restatement's normals (as float32) up to the
    generator's 4e-6 per draw."""
    from bsi_amd import BSI, Discretization

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.5))

        def forward(self, x, t):
            return self.w * x

    shape, k, n = (3, 8, 8), 4, 4
    bsi = BSI(Net(), data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k, preconditioning="edm",
              discretization=Discretization.image_8bit()).to(DEV)
    g = torch.Generator(DEV).manual_seed(3)
    with torch.no_grad():
        s = bsi.sample(n, g, device_noise=True)
        g2 = torch.Generator(DEV).manual_seed(3)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=DEV, generator=g2))
        cnt = n * 3 * 8 * 8
        noise0 = torch.from_numpy(po.normal_stream(seed, po.MU0_STREAM, cnt).astype(np.float32)).to(DEV).view(n, *shape)
        steps = torch.stack([torch.from_numpy(po.normal_stream(seed, i, cnt).astype(np.float32)).view(n, *shape) for i in range(k)]).to(DEV)
        ref = bsi._run_chain(n, None, None, history=False, noise=(noise0, steps))
    err = float((s - ref).abs().max())
    report("device_noise_chain_vs_restatement", max_abs=err)
    assert err < 1e-4  # outputs in [-1, 1]; per-draw deviation 4e-6 amplified by at most sqrt(alpha)/lambda ratios of the chain
