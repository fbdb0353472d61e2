"""GPU parity tests of the VDM and BFN wrappers (bsi_amd/vdm.py, bsi_amd/bfn.py) on the HIP path against the golden vectors
generated from the reference's own classes (tests/golden/g9_vdm_*, g10_bfn_*) and the CPU oracles.  The wrapper arithmetic is
fp32 (tolerances 1e-5..1e-4 where the denoiser is not involved); everything that goes through the bf16 denoiser uses the
tolerances of tests/test_hip_dit.py (1e-3 on losses, 1e-2 on one-step predictions and on the relative L2 error of every gradient tensor)."""
import pytest
import torch

from tests.test_hip_dit import make_model, replay_noise
from tests.util import bound, golden, max_rel, rel_linf

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPE = (3, 16, 16)


def make_vdm(model, k=8):
    from bsi_amd import VDM, Discretization
    return VDM(model, data_shape=SHAPE, snr_min=6.73794699909e-3, snr_max=597195.613793, k=k,
               discretization=Discretization.image_8bit()).to(DEV)


def make_bfn(model, k=8):
    from bsi_amd import BFN, Discretization
    return BFN(model, data_shape=SHAPE, sigma_1=1e-3, k=k, discretization=Discretization.image_8bit()).to(DEV)


def _check_grads(model, g, tol=1e-2, tag="grads"):
    for name, p in model.named_parameters():
        ref = g["G." + name]
        assert p.grad is not None, name
        err = float((p.grad.cpu().double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))
        bound(tag, err, tol)


def test_vdm_schedule_functions_and_coefficients():
    v = make_vdm(make_model())
    g = golden("g9_vdm_tables")
    t = g["t"].to(DEV)
    for name in ("sigma2", "alpha", "snr"):
        bound("test_vdm_schedule_functions_and_coefficients:40", max_rel(getattr(v, name)(t), g[name]), 2e-6)
    bound("test_vdm_schedule_functions_and_coefficients:41", max_rel(v.gamma(t), g["gamma"]), 1e-6)
    with pytest.raises(NotImplementedError):
        v.diffusion_loss(t, 1)


def test_vdm_losses_vs_golden():
    g = golden("g9_vdm_train")
    model = make_model()
    v = make_vdm(model)
    with replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps"]]):
        loss = v.train_loss(g["x"].to(DEV))
    assert loss.shape == g["loss"].shape and max_rel(loss.detach(), g["loss"]) < 1e-2
    loss.mean().backward()
    _check_grads(model, g)
    g = golden("g9_vdm_elbo")
    v = make_vdm(make_model())
    with torch.no_grad(), replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps_recon"], g["eps_diff"]]):
        elbo, bpd, extra = v.elbo(g["x"].to(DEV), 2, 2, estimate_var=True)
    bound("test_vdm_losses_vs_golden:59", max_rel(extra["l_prior"], g["l_prior"]), 1e-5)
    assert max_rel(extra["l_recon"], g["l_recon"]) < 1e-4   # no denoiser involved: fp32 wrapper arithmetic only
    bound("test_vdm_losses_vs_golden:61", max_rel(extra["l_diff"], g["l_diff"]), 1e-3)
    bound("test_vdm_losses_vs_golden:62a", max_rel(bpd, g["bpd"]), 1e-3)
    bound("test_vdm_losses_vs_golden:62b", max_rel(elbo, g["elbo"]), 1e-3)
    bound("test_vdm_losses_vs_golden:63", rel_linf(extra["bpd_var"], g["bpd_var"]), 1e-2)
    g = golden("g9_vdm_finite_elbo")
    with torch.no_grad(), replay_noise(randint=[g["i"]], randn=[g["eps_recon"], g["eps_diff"]]):
        elbo, bpd, extra = v.finite_elbo(g["x"].to(DEV), 2, 2)
    bound("test_vdm_losses_vs_golden:67a", max_rel(extra["l_recon"], g["l_recon"]), 1e-4)
    bound("test_vdm_losses_vs_golden:67b", max_rel(extra["l_diff"], g["l_diff"]), 1e-3)
    bound("test_vdm_losses_vs_golden:68", max_rel(bpd, g["bpd"]), 1e-3)
    with pytest.raises(AssertionError):
        v.elbo(g["x"].to(DEV), 1, 2, estimate_var=True)


def test_vdm_sampler_teacher_forced_and_free_running():
    g = golden("g9_vdm_hist")
    v = make_vdm(make_model(), k=int(g["k"]))
    k = int(g["k"])
    ts = torch.linspace(1.0, 0.0, k + 1, device=DEV)
    with torch.no_grad():
        for i in range(k):  # one ancestral step from the reference's z_t: x_hat and z_s
            z = g["zs"][i].to(DEV)
            xh = v._predict_x(z, ts[i].repeat(len(z)))
            bound("test_vdm_sampler_teacher_forced_and_free_running:82", rel_linf(xh, g["x_hats"][i]), 1e-2)
            with replay_noise(randn=[g["eps"][i]]):
                zs = v._sample_zs_given_zt_x(ts[i + 1].repeat(len(z)), z, ts[i].repeat(len(z)), g["x_hats"][i].to(DEV))
            bound("test_vdm_sampler_teacher_forced_and_free_running:85", rel_linf(zs, g["zs"][i + 1]), 1e-5)
        with replay_noise(randn=[g["eps0"], *g["eps"]]):
            x_hats = v.sample_history(2)
        with replay_noise(randn=[g["eps0"], *g["eps"]]):
            smp = v.sample(2)
    assert torch.equal(smp, x_hats[-1]) and torch.isfinite(smp).all()
    assert rel_linf(x_hats[0], g["x_hats"][0]) < 2e-2   # first step identical inputs; later steps amplify bf16 differences
    bound("test_vdm_sampler_teacher_forced_and_free_running:92", rel_linf(x_hats, g["x_hats"]), 0.25)


def test_bfn_losses_vs_golden():
    g = golden("g10_bfn_train")
    model = make_model()
    b = make_bfn(model)
    with replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps"]]):
        loss = b.train_loss(g["x"].to(DEV))
    assert loss.ndim == 0 and abs(float(loss.detach()) / float(g["loss"]) - 1) < 1e-2
    loss.backward()
    _check_grads(model, g)
    g = golden("g10_bfn_elbo")
    b = make_bfn(make_model())
    with torch.no_grad(), replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps_recon"], g["eps_latent"]]):
        elbo, bpd, extra = b.elbo(g["x"].to(DEV), 2, 2, estimate_var=True)
    bound("test_bfn_losses_vs_golden:108a", max_rel(extra["l_recon"], g["l_recon"]), 1e-2)
    bound("test_bfn_losses_vs_golden:108b", max_rel(extra["l_latent"], g["l_latent"]), 1e-3)
    bound("test_bfn_losses_vs_golden:109a", max_rel(bpd, g["bpd"]), 1e-2)
    bound("test_bfn_losses_vs_golden:109b", rel_linf(extra["bpd_var"], g["bpd_var"]), 1e-2)
    g = golden("g10_bfn_finite_elbo")
    with torch.no_grad(), replay_noise(randint=[g["i"]], randn=[g["eps_recon"], g["eps_latent"]]):
        elbo, bpd, extra = b.finite_elbo(g["x"].to(DEV), 2, 2, t=g["t"].to(DEV))
    bound("test_bfn_losses_vs_golden:113a", max_rel(extra["l_latent"], g["l_latent"]), 1e-2)
    bound("test_bfn_losses_vs_golden:113b", max_rel(bpd, g["bpd"]), 1e-3)
    with pytest.raises(AttributeError):   # the reference's `self.linspace` quirk
        b.finite_elbo(g["x"].to(DEV), 2, 2)


def test_bfn_predict_and_sampler():
    b = make_bfn(make_model())
    g = golden("g10_bfn_predict")
    with torch.no_grad():
        xh = b._predict_x(g["mu"].to(DEV), g["t"].to(DEV))
    assert float(xh[:2].abs().max()) == 0.0 and float(xh.abs().max()) <= 1.0   # t < t_min -> 0; clipped to [x_min, x_max]
    # row 2 (t = 1e-3): gamma = 0.0137, the bf16 denoiser's error enters multiplied by sqrt((1 - gamma) / gamma) = 8.5
    bound("test_bfn_predict_and_sampler:125a", rel_linf(xh[3], g["x_hat"][3]), 2e-2)
    bound("test_bfn_predict_and_sampler:125b", rel_linf(xh, g["x_hat"]), 5e-2)
    g = golden("g10_bfn_hist")
    k = int(g["k"])
    b = make_bfn(make_model(), k=k)
    t = torch.linspace(0, 1, k + 1, device=DEV)
    alpha, rho, _ = b._schedule(t)
    with torch.no_grad():
        for i in range(k + 1):  # teacher forced one-step predictions
            mu = g["mus"][i].to(DEV)
            ti = t[i] if i < k else t.new_ones(())
            xh = b._predict_x(mu, ti.repeat(len(mu)))
            bound("test_bfn_predict_and_sampler:136", rel_linf(xh, g["x_hats"][i]), 1e-2)
        with replay_noise(randn=list(g["eps"])):
            mus, x_hats, ys = b.sample_history(2)
        with replay_noise(randn=list(g["eps"])):
            smp = b.sample(2)
    assert torch.equal(smp, x_hats[-1]) and torch.isfinite(mus).all()
    # rho_{i+1} = rho_i + alpha_i, rho_0 = 1: posterior precision of the refine update
    assert abs(float(rho[0]) - 1) < 1e-7 and rel_linf(rho[1:] - rho[:-1], alpha) < 1e-5
    bound("test_bfn_predict_and_sampler:144a", rel_linf(x_hats[:2], g["x_hats"][:2]), 2e-2)
    bound("test_bfn_predict_and_sampler:144b", rel_linf(mus[:2], g["mus"][:2]), 1e-5)
    bound("test_bfn_predict_and_sampler:145", rel_linf(x_hats, g["x_hats"]), 0.3)
