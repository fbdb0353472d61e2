"""The CPU restatements of the VDM and BFN wrappers (oracle/vdm_oracle.py, oracle/bfn_oracle.py) against golden vectors
produced by the reference's own classes (tools/gen_golden_algos.py).  fp32 CPU on both sides: agreement to rounding."""
import torch

from oracle import bsi_oracle as bo
from oracle import dit_oracle as do
from oracle.bfn_oracle import BFNOracle
from oracle.vdm_oracle import VDMOracle
from tests.util import golden, max_rel, rel_linf, weights

SHAPE = (3, 16, 16)


def _f():
    W = weights("dit_ff")
    return lambda mu, t: do.dit_forward(W, mu, t, patch_size=2, dim=128, depth=2, heads=2, ff=(6, 8))


def _vdm(k=8):
    return VDMOracle(_f(), data_shape=SHAPE, k=k, discretization=bo.Disc.image_8bit())


def _bfn(k=8):
    return BFNOracle(_f(), data_shape=SHAPE, sigma_1=1e-3, k=k, discretization=bo.Disc.image_8bit())


def test_vdm_tables_and_losses():
    o = _vdm()
    g = golden("g9_vdm_tables")
    for name in ("gamma", "sigma2", "alpha", "snr"):
        assert max_rel(getattr(o, name)(g["t"]), g[name]) < 1e-6, name
    g = golden("g9_vdm_train")
    with torch.no_grad():
        assert max_rel(o.train_loss(g["x"], g["offset"], g["perm"], g["eps"][0]), g["loss"]) < 1e-5
    g = golden("g9_vdm_elbo")
    with torch.no_grad():
        assert max_rel(o.prior_loss(g["x"]), g["l_prior"]) < 1e-6
        assert max_rel(o.reconstruction_loss(g["x"], g["eps_recon"]), g["l_recon"]) < 1e-5
        assert max_rel(o.inf_diffusion_loss(g["x"], g["offset"], g["perm"], g["eps_diff"]), g["l_diff"]) < 1e-4
    g = golden("g9_vdm_finite_elbo")
    with torch.no_grad():
        assert max_rel(o.reconstruction_loss(g["x"], g["eps_recon"]), g["l_recon"]) < 1e-5
        assert max_rel(o.finite_diffusion_loss(g["x"], g["i"], g["eps_diff"]), g["l_diff"]) < 1e-4


def test_vdm_sampler():
    o = _vdm()
    g = golden("g9_vdm_hist")
    with torch.no_grad():
        xs, zs = o.sample_history(g["eps0"], g["eps"], teacher_z=g["zs"])  # every step starts from the reference's z_t
    assert rel_linf(zs, g["zs"]) < 1e-4 and rel_linf(xs[:-1], g["x_hats"][:-1]) < 1e-4
    with torch.no_grad():
        xs_free, _ = o.sample_history(g["eps0"], g["eps"])  # free-running: fp32 differences of the denoiser get amplified
    assert rel_linf(xs_free, g["x_hats"]) < 2e-2


def test_bfn_losses_and_sampler():
    o = _bfn()
    g = golden("g10_bfn_train")
    with torch.no_grad():
        assert max_rel(o.train_loss(g["x"], g["offset"], g["perm"], g["eps"]), g["loss"]) < 1e-5
    g = golden("g10_bfn_elbo")
    with torch.no_grad():
        assert max_rel(o.reconstruction_loss(g["x"], g["eps_recon"]), g["l_recon"]) < 1e-5
        assert max_rel(o.continuous_time_loss(g["x"], g["offset"], g["perm"], g["eps_latent"]), g["l_latent"]) < 1e-4
    g = golden("g10_bfn_finite_elbo")
    with torch.no_grad():
        assert max_rel(o.discrete_time_loss(g["x"], g["i"], g["eps_latent"], g["t"]), g["l_latent"]) < 1e-4
    g = golden("g10_bfn_hist")
    with torch.no_grad():
        mus, xs, ys = o.sample_history(2, g["eps"], teacher_mus=g["mus"])  # teacher forced
    assert rel_linf(mus, g["mus"]) < 1e-4 and rel_linf(xs, g["x_hats"]) < 1e-4 and rel_linf(ys, g["ys"]) < 1e-4
    with torch.no_grad():
        mus_f, xs_f, _ = o.sample_history(2, g["eps"])
    assert rel_linf(xs_f, g["x_hats"]) < 5e-2
    g = golden("g10_bfn_predict")
    with torch.no_grad():
        xh = o.predict_x(g["mu"], g["t"])
    assert rel_linf(xh, g["x_hat"]) < 1e-5 and float(xh[:2].abs().max()) == 0.0 and float(xh.abs().max()) <= 1.0
