"""Pin the CPU oracle (oracle/) against the golden vectors generated from the reference
(tools/gen_golden.py) and against the reference's own four known-answer tests.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import bsi_oracle as bo
from oracle import dit_oracle as do
from oracle import unet_oracle as uo
from tests.util import golden, max_rel, rel_linf, sub, weights

F32 = torch.float32


def tiny_conv(W):
    """README.md:21-28 denoiser as a function."""
    def f(mu, t):
        tp = t.reshape(-1, 1, 1, 1).expand(-1, 1, *mu.shape[-2:])
        return torch.nn.functional.conv2d(torch.cat((mu, tp), dim=1), W["layer.weight"], W["layer.bias"], padding=1)
    return f


def dit_f(W, ff=True, md=None):
    dim = W["dit.patch_encoder.weight"].shape[0]
    depth = 1 + max(int(k.split(".")[2]) for k in W if k.startswith("dit.blocks."))
    return lambda mu, t: do.dit_forward(W, mu, t, patch_size=2, dim=dim, depth=depth, heads=dim // 64,
                                        ff=(6, 8) if ff else None, md=md)


def unet_f(W, ff=True, md=None):
    levels = 1 + max(int(k.split(".")[2]) for k in W if k.startswith("u_net.downsampling_blocks."))
    slot = any(".layers.6." in k for k in W)
    return lambda mu, t: uo.unet_forward(W, mu, t, levels=levels, ff=(6, 8) if ff else None,
                                         has_dropout_slot=slot, md=md)


def make(f, shape, k=16, disc=True, dtype=F32):
    return bo.BSIOracle(f, data_shape=shape, k=k, dtype=dtype,
                        discretization=bo.Disc.image_8bit() if disc else None)


# ----------------------------------------------------------------------------------------
# The reference's own known-answer tests (tests/test_bsi.py, test_fourier_features.py)
# ----------------------------------------------------------------------------------------
def test_kat_bucketize_rgb():
    d = bo.Disc(0.0, 1.0, 256)
    x = torch.tensor([-0.1, 0.0, 1.0, 1.0 - 1 / 256], dtype=torch.float64)
    assert d.bucketize(x).tolist() == [0, 0, 255, 254]
    g = golden("kat_reference_tests")
    assert torch.equal(d.bucketize(g["x1"]), g["idx1"])


def test_kat_bucketize_aligns_with_boundaries():
    d = bo.Disc(-1.0, 1.0, 5)
    b = d.bin_boundaries(torch.float64)
    assert d.bucketize(b)[:-1].tolist() == list(range(5))
    assert d.bucketize(b - 1e-8)[1:].tolist() == list(range(5))


def test_kat_bin_boundaries():
    d = bo.Disc(-1.0, 1.0, 3)
    np.testing.assert_allclose(d.bin_boundaries(F32), [-1.5, -0.5, 0.5, 1.5])


def test_kat_fourier_features():
    x = torch.tensor([1.333, -np.e / 7], dtype=torch.float64)[None, :, None].repeat(2, 1, 3)
    y = do.fourier_features(x, 5, 6, dim=1, table_dtype=torch.float64)  # conftest: double default
    assert y.shape == (2, 8, 3)
    exp = [f(2 * np.pi * 2 ** n * v) for v in (1.333, -np.e / 7) for n in (5, 6) for f in (np.sin, np.cos)]
    np.testing.assert_allclose(y[0, :, 0], exp, rtol=1e-7, atol=1e-9)
    g = golden("kat_reference_tests")
    np.testing.assert_allclose(y, g["ff_y"], rtol=1e-12, atol=1e-12)
    assert torch.equal(bo.Disc.image_8bit().to_8bit_image(torch.tensor([-1.2, -1.0, -0.5, 0.0, 0.999, 1.0, 1.5])),
                       g["img8"])


# ----------------------------------------------------------------------------------------
# G1-G3: schedule tables, lambda grids, forward-process samples
# ----------------------------------------------------------------------------------------
def test_g1_tables():
    g = golden("g1_tables")
    o = make(None, (3, 8, 8))
    assert o.p_lambda.ln_low == float(g["ln_low"]) and o.p_lambda.delta == float(g["delta"])
    assert abs(o.p_lambda.ln_low - (-4.605170208339834)) < 1e-15
    assert abs(o.p_lambda.delta - 18.42068076630411) < 1e-13
    lam = o.p_lambda.icdf(g["t"])
    assert torch.equal(lam, g["lam"])
    cs, co, ci = o.edm_coeffs(g["t"])
    for a, b in [(cs, g["c_skip"]), (co, g["c_out"]), (ci, g["c_in"]),
                 (o.p_lambda.cdf(lam), g["cdf_lam"]), (o.p_lambda.reciprocal_pdf(lam), g["rpdf"])]:
        assert torch.equal(a, b)
    assert torch.equal(o.default_schedule, g["default_schedule"])
    o64 = make(None, (3, 8, 8), dtype=torch.float64)
    assert torch.equal(o64.p_lambda.icdf(g["t"].double()), g["lam64"])
    cs, co, ci = o64.edm_coeffs(g["t"].double())
    assert torch.equal(cs, g["c_skip64"]) and torch.equal(co, g["c_out64"]) and torch.equal(ci, g["c_in64"])


def test_g2_g3_lambda_grid_and_q():
    g = golden("g2g3_lambda_q")
    o = make(None, (3, 8, 8))
    for n, B in [(1, 8), (3, 5)]:
        lam = o.lambda_grid(g[f"offset_{n}_{B}"], g[f"perm_{n}_{B}"], n, B)
        assert torch.equal(lam, g[f"lam_{n}_{B}"])
    assert torch.equal(o.q_mu_lambda(g["x"], g["q_lam"], g["q_eps"]), g["q_mu"])


# ----------------------------------------------------------------------------------------
# G4: train_loss (+ gradients through the oracle by autograd)
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,shape,builder,wtag", [
    ("g4_train_tinyconv", (3, 8, 8), tiny_conv, None),
    ("g4_train_dit", (3, 16, 16), lambda W: dit_f(W, True), "dit_ff"),
    ("g4_train_dit_noff", (3, 16, 16), lambda W: dit_f(W, False), "dit_noff"),
    ("g4_train_unet", (3, 8, 8), lambda W: unet_f(W, True), "unet_ff"),
])
def test_g4_train_loss_and_grads(case, shape, builder, wtag):
    g = golden(case)
    W = weights(wtag) if wtag else sub(g, "W.")
    W = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in W.items()}
    o = make(builder(W), shape)
    loss = o.train_loss(g["x"], g["offset"], g["perm"], g["eps"])
    assert max_rel(loss, g["loss"]) < 2e-5, max_rel(loss, g["loss"])
    assert abs(float(loss.mean()) / float(g["loss_mean"]) - 1) < 1e-5
    loss.mean().backward()
    G = sub(g, "G.")
    sq = 0.0
    for k, ref in G.items():
        got = W[k].grad
        sq += float((got.double() ** 2).sum())
        assert rel_linf(got, ref) < 2e-3, (k, rel_linf(got, ref))
    assert abs(math.sqrt(sq) / float(g["grad_norm"]) - 1) < 1e-4


def test_g11_calibration_point():
    """SURVEY Appendix F calibration model (dim 128, depth 4, heads 2, patch 4, B = 64): the fp32 oracle against the
    reference's fp32 and fp64 train_loss, per-tensor gradient norms and teacher-forced predictions."""
    from tests.util import CALIB as c, calib_weights
    g = golden("g11_calib_dit")
    W = {k: v.clone().requires_grad_(True) for k, v in calib_weights().items()}
    f = lambda mu, t: do.dit_forward(W, mu, t, patch_size=c["patch_size"], dim=c["dim"], depth=c["depth"],  # noqa: E731
                                     heads=c["heads"], ff=c["ff"])
    o = make(f, c["shape"], k=128)
    loss = o.train_loss(g["x"], g["offset"], g["perm"], g["eps"])
    assert max_rel(loss, g["loss"]) < 2e-5, max_rel(loss, g["loss"])
    assert max_rel(loss, g["loss_fp64"]) < 2e-5
    assert abs(float(loss.mean()) / float(g["loss_mean"]) - 1) < 1e-5
    loss.mean().backward()
    sq = 0.0
    for k, v in W.items():
        n = float(v.grad.double().norm())
        sq += n * n
        assert abs(n - float(g["GN." + k])) <= 2e-3 * float(g["GN." + k]) + 1e-7 * float(g["grad_norm"]), k
    assert abs(math.sqrt(sq) / float(g["grad_norm"]) - 1) < 1e-4
    with torch.no_grad():
        xh = o.predict_x(g["tf_mu"], g["tf_t"])
    assert rel_linf(xh, g["tf_xhat"]) < 1e-4 and rel_linf(xh, g["tf_xhat64"]) < 1e-4


def test_g13_unet_calibration_point():
    """UNet calibration model (dim 128, 2 levels, 16x16, B = 64): the fp32 oracle against the reference's fp32 and fp64
    train_loss, per-tensor gradient norms and teacher-forced predictions."""
    from tests.util import CALIB_UNET as c, calib_unet_weights
    g = golden("g13_calib_unet")
    W = {k: v.clone().requires_grad_(True) for k, v in calib_unet_weights().items()}
    f = lambda mu, t: uo.unet_forward(W, mu, t, levels=c["levels"], ff=c["ff"], has_dropout_slot=True)  # noqa: E731
    o = make(f, c["shape"], k=128)
    loss = o.train_loss(g["x"], g["offset"], g["perm"], g["eps"])
    assert max_rel(loss, g["loss"]) < 2e-5, max_rel(loss, g["loss"])
    assert max_rel(loss, g["loss_fp64"]) < 2e-5
    loss.mean().backward()
    sq = 0.0
    for k, v in W.items():
        n = float(v.grad.double().norm())
        sq += n * n
        assert abs(n - float(g["GN." + k])) <= 2e-3 * float(g["GN." + k]) + 1e-7 * float(g["grad_norm"]), k
    assert abs(math.sqrt(sq) / float(g["grad_norm"]) - 1) < 1e-4
    with torch.no_grad():
        xh = o.predict_x(g["tf_mu"], g["tf_t"])
    assert rel_linf(xh, g["tf_xhat"]) < 1e-4 and rel_linf(xh, g["tf_xhat64"]) < 1e-4


def test_g4_fp64_oracle_matches_fp64_reference():
    g = golden("g4_train_dit")
    W = {k: v.double() for k, v in weights("dit_ff").items()}
    o = make(dit_f(W, True), (3, 16, 16), dtype=torch.float64)
    # same noise in double (the fp64 reference run used .double() of the fp32 draws)
    lam = o.p_lambda.icdf(torch.remainder(g["perm"].double() / (1 + 4) + g["offset"].double(), 1))
    mu = o.q_mu_lambda(g["x"].double(), lam, g["eps"].double())
    xh = o.predict_x(mu, o.p_lambda.cdf(lam))
    loss = o.p_lambda.reciprocal_pdf(lam) * (g["x"].double() - xh).square().flatten(1).mean(1)
    assert max_rel(loss, g["loss_fp64"]) < 1e-9


# ----------------------------------------------------------------------------------------
# G5: sampling trajectories — free-running without Fourier features, teacher-forced with
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,shape,builder,wtag,teacher", [
    ("g5_hist_tinyconv", (3, 8, 8), tiny_conv, None, False),
    ("g5_hist_dit_noff", (3, 16, 16), lambda W: dit_f(W, False), "dit_noff", False),
    ("g5_hist_unet_noff", (3, 8, 8), lambda W: unet_f(W, False), "unet_noff", False),
    ("g5_hist_dit_ff", (3, 16, 16), lambda W: dit_f(W, True), "dit_ff", True),
    ("g5_hist_unet_ff", (3, 8, 8), lambda W: unet_f(W, True), "unet_ff", True),
])
def test_g5_sample_history(case, shape, builder, wtag, teacher):
    g = golden(case)
    W = weights(wtag) if wtag else sub(g, "W.")
    o = make(builder(W), shape, k=int(g["k"]))
    with torch.no_grad():
        mus, xh, ys = o.sample_history(g["eps0"], g["eps"], teacher_mus=g["mus"] if teacher else None)
    tol = 1e-4 if teacher else 2e-5
    assert mus.shape == g["mus"].shape and xh.shape == g["x_hats"].shape and ys.shape == g["ys"].shape
    for a, b, n in [(mus, g["mus"], "mu"), (xh, g["x_hats"], "x_hat"), (ys, g["ys"], "y")]:
        for i in range(a.shape[0]):
            assert rel_linf(a[i], b[i]) < tol, (n, i, rel_linf(a[i], b[i]))


# ----------------------------------------------------------------------------------------
# G6: ELBO pieces
# ----------------------------------------------------------------------------------------
def test_g6_elbo():
    g = golden("g6_elbo")
    W = sub(g, "W.")
    o = make(tiny_conv(W), (3, 8, 8), k=16)
    with torch.no_grad():
        lr = o.reconstruction_loss(g["x"], g["eps_r"])
        lm = o.inf_measurement_loss(g["x"], g["offset"], g["perm"], g["eps_m"])
        elbo, bpd, extra = o.assemble_elbo(lr, lm, estimate_var=True)
        assert max_rel(lr, g["l_recon"]) < 1e-5 and max_rel(lm, g["l_measure"]) < 1e-5
        assert max_rel(elbo, g["elbo"]) < 1e-5 and max_rel(bpd, g["bpd"]) < 1e-5
        assert max_rel(extra["bpd_var"], g["bpd_var"]) < 1e-4
        flr = o.reconstruction_loss(g["x"], g["feps_r"])
        flm = o.finite_measurement_loss(g["x"], g["fidx"], g["feps_m"], t=torch.linspace(0, 1, 17))
        felbo, fbpd, fextra = o.assemble_elbo(flr, flm, estimate_var=True)
        assert max_rel(flm, g["fl_measure"]) < 1e-5 and max_rel(felbo, g["felbo"]) < 1e-5
        assert max_rel(fextra["bpd_var"], g["fbpd_var"]) < 1e-4
        oc = make(tiny_conv(W), (3, 8, 8), k=16, disc=False)
        assert max_rel(oc.reconstruction_loss(g["x"], g["ceps_r"]), g["cl_recon"]) < 1e-5
    with pytest.raises(AssertionError):
        o.assemble_elbo(lr[:1], lm, estimate_var=True)


# ----------------------------------------------------------------------------------------
# G15: the two branches of the surface nothing else covers -- preconditioning=None (bsi.py:379-380) and
# low_discrepancy_sampling=False (bsi.py:441-445, the transposed (batch, n) grid)
# ----------------------------------------------------------------------------------------
def test_g15_no_preconditioning():
    g = golden("g15_branches")
    W = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in weights("dit_noff").items()}
    o = bo.BSIOracle(dit_f(W, False), data_shape=(3, 16, 16), k=4, preconditioning=None, discretization=bo.Disc.image_8bit())
    loss = o.train_loss(g["A_x"], g["A_offset"], g["A_perm"], g["A_eps"])
    assert max_rel(loss, g["A_loss"]) < 2e-5 and abs(float(loss.mean()) / float(g["A_loss_mean"]) - 1) < 1e-5
    loss.mean().backward()
    for k, ref in sub(g, "A_G.").items():
        assert rel_linf(W[k].grad, ref) < 2e-3, (k, rel_linf(W[k].grad, ref))
    with torch.no_grad():
        mus, xh, ys = o.sample_history(g["A_eps0"], g["A_eps_steps"])
        for a, b in [(mus, g["A_mus"]), (xh, g["A_x_hats"]), (ys, g["A_ys"])]:
            assert a.shape == b.shape
            for i in range(a.shape[0]):
                assert rel_linf(a[i], b[i]) < 2e-5, (i, rel_linf(a[i], b[i]))
        lr = o.reconstruction_loss(g["A_x"], g["A_eps_r"])
        lm = o.inf_measurement_loss(g["A_x"], g["A_e_offset"], g["A_e_perm"], g["A_eps_m"])
        elbo, bpd, extra = o.assemble_elbo(lr, lm, estimate_var=True)
        assert max_rel(lr, g["A_l_recon"]) < 1e-5 and max_rel(lm, g["A_l_measure"]) < 2e-5
        assert max_rel(elbo, g["A_elbo"]) < 1e-5 and max_rel(bpd, g["A_bpd"]) < 1e-5
        assert max_rel(extra["bpd_var"], g["A_bpd_var"], floor=1e-12) < 1e-3


def test_g15_plain_lambda_sampling():
    g = golden("g15_branches")
    o = make(dit_f(weights("dit_noff"), False), (3, 16, 16), k=4)
    lam = o.lambda_plain(g["B_u"])
    assert lam.shape == (5, 3) and torch.equal(lam, g["B_lam"])  # (batch, n): the reference's transposed shape
    with torch.no_grad():
        loss = o.train_loss_plain(g["A_x"], g["B_t_u"], g["B_t_eps"])
        assert loss.shape == (4,) and max_rel(loss, g["B_t_loss"]) < 2e-5
        ot = make(tiny_conv(sub(g, "B_m_W.")), (3, 8, 8), k=4)
        lm = ot.inf_measurement_loss_plain(g["B_m_x"], g["B_m_u"], g["B_m_eps"])
        assert lm.shape == g["B_m_loss"].shape and max_rel(lm, g["B_m_loss"]) < 1e-5


# ----------------------------------------------------------------------------------------
# G7: components and full forwards
# ----------------------------------------------------------------------------------------
def test_g7_components():
    g = golden("g7_components")
    for size, rate in [(1024, 1000), (32, 100), (512, 32), (64, 16)]:
        sc, bi = do.nyquist_tables(size, rate)
        assert torch.equal(sc, g[f"pe_{size}_{rate}_scale"]) and torch.equal(bi, g[f"pe_{size}_{rate}_bias"])
        out = do.nyquist_embedding(g[f"pe_{size}_{rate}_t"], size, rate)
        assert torch.equal(out, g[f"pe_{size}_{rate}_out"])
    assert torch.equal(do.fourier_features(g["ff_x"], 6, 8), g["ff_out"])
    assert torch.equal(do.patch_pos_embedding(128, 16, 16, 2), g["dit16_pos"])
    B = sub(g, "BLK.")
    out = do.dit_block(g["blk_x"], g["blk_c"], B, "", heads=2)
    assert rel_linf(out, g["blk_out"]) < 1e-5
    att = do.attention(g["blk_x"], B["attn.to_qkv.weight"], B["attn.to_qkv.bias"],
                       B["attn.to_out.weight"], B["attn.to_out.bias"], 2)
    assert rel_linf(att, g["attn_out"]) < 1e-5
    for tag in ("rb64", "rb128"):
        R = sub(g, tag.upper() + ".")
        out = uo.residual_block(g[f"{tag}_x"], g[f"{tag}_c"], R, "", has_dropout_slot=True)
        assert rel_linf(out, g[f"{tag}_out"]) < 1e-5
    A = sub(g, "A2D.")
    assert rel_linf(uo.attention2d(g["a2d_x"], A, "", 1), g["a2d_out"]) < 1e-5


def test_g7_full_forwards():
    g = golden("g7_dit_fwd")
    W = weights("dit_ff")
    with torch.no_grad():
        y = dit_f(W, True)(g["mu"], g["t"])
    assert rel_linf(y, g["out"]) < 2e-5
    W64 = {k: v.double() for k, v in W.items()}
    with torch.no_grad():
        y64 = dit_f(W64, True)(g["mu"].double(), g["t"].double())
    assert rel_linf(y64, g["out64"]) < 1e-10
    g = golden("g7_unet_fwd")
    W = weights("unet_ff")
    with torch.no_grad():
        y = unet_f(W, True)(g["mu"], g["t"])
    assert rel_linf(y, g["out"]) < 2e-5
    assert set(W) == set(uo.unet_param_shapes((3, 8, 8), 64, 1, ff=(6, 8)))
    assert all(tuple(W[k].shape) == s for k, s in uo.unet_param_shapes((3, 8, 8), 64, 1, ff=(6, 8)).items())
    Wd = weights("dit_ff")
    shp = do.dit_param_shapes((3, 16, 16), 2, 128, 2, ff=(6, 8))
    assert set(Wd) == set(shp) and all(tuple(Wd[k].shape) == s for k, s in shp.items())


# ----------------------------------------------------------------------------------------
# G8: EMA schedule and clip + AdamW
# ----------------------------------------------------------------------------------------
def test_g8_ema_and_adamw():
    g = golden("g8_optimizer")
    steps, dec = g["ema_steps"].tolist(), g["ema_decays"].tolist()
    for s, d in zip(steps, dec):
        # reference: step is incremented before get_current_decay() is evaluated
        assert abs(bo.ema_decay(s + 1) - d) < 1e-15, (s, d)
    assert bo.ema_decay(1001) == 0.0 and abs(bo.ema_decay(1002) - (1 - 2 ** (-2 / 3))) < 1e-12
    names = sorted(k[3:] for k in g if k.startswith("p0."))
    P = [g["p0." + n].clone() for n in names]
    M = [torch.zeros_like(p) for p in P]
    V = [torch.zeros_like(p) for p in P]
    for step in (1, 2, 3):
        G = [g[f"g{step}.{n}"] for n in names]
        norm = bo.clip_adamw_step(P, G, M, V, step, lr=5e-4, beta1=0.9, beta2=0.99, eps=1e-8,
                                  weight_decay=1e-2, max_norm=1.0)
        assert abs(float(norm) / float(g[f"norm{step}"]) - 1) < 1e-6
        for p, n in zip(P, names):
            assert rel_linf(p, g[f"p{step}.{n}"]) < 1e-6, (step, n)
    a = g["lerp_tgt"].clone()
    a = a + (g["lerp_src"] - a) * (1.0 - 0.9)
    assert rel_linf(a, g["lerp_out_0.9"]) < 1e-6


# ----------------------------------------------------------------------------------------
# G14: BASELINE.json configs[0] at its stated size (README denoiser, 3x32x32, batch 32, k = 16)
# ----------------------------------------------------------------------------------------
def config1_noise(g):
    """Re-draw the Gaussian noise of g14_config1 from its seeds (same generator calls as the reference made, bsi.py:232-247,
    325-334) and check the fingerprints the generator script stored."""
    B, shape, k = 32, (3, 32, 32), int(g["k"])

    def fp(t):
        d = t.double()
        return torch.stack((d.sum(), d.abs().sum(), d.flatten()[0], d.flatten()[-1], (d * d).sum()))

    gen = torch.Generator().manual_seed(int(g["seed_train"]))
    off, perm = torch.rand((), generator=gen), torch.randperm(B, generator=gen)
    eps = torch.randn((B, *shape), generator=gen)
    def same(a, b):  # sums are taken in a thread-count dependent order: 1e-12 relative; first / last element exact
        return torch.allclose(a, b, rtol=1e-12, atol=0) and torch.equal(a[2:4], b[2:4])

    assert torch.equal(off, g["offset"]) and torch.equal(perm, g["perm"]) and same(fp(eps), g["eps_fp"]), \
        "torch CPU generator stream differs from the one the golden was drawn with"
    gen = torch.Generator().manual_seed(int(g["seed_sample"]))
    eps0 = torch.randn((B, *shape), generator=gen)
    eps_s = torch.stack([torch.randn((B, *shape), generator=gen) for _ in range(k)])
    assert same(fp(eps0), g["eps0_fp"]) and same(fp(eps_s), g["eps_steps_fp"])
    return off, perm, eps, eps0, eps_s


def test_g14_config1_at_stated_size():
    g = golden("g14_config1")
    off, perm, eps, eps0, eps_s = config1_noise(g)
    W = {k: v.clone().requires_grad_(True) for k, v in sub(g, "W.").items()}
    o = make(tiny_conv(W), (3, 32, 32), k=int(g["k"]))
    loss = o.train_loss(g["x"], off, perm, eps)
    assert max_rel(loss, g["loss"]) < 1e-5 and abs(float(loss.mean()) / float(g["loss_mean"]) - 1) < 1e-6
    loss.mean().backward()
    for k, ref in sub(g, "G.").items():
        assert rel_linf(W[k].grad, ref) < 1e-4, (k, rel_linf(W[k].grad, ref))
    with torch.no_grad():
        mus, xh, ys = o.sample_history(eps0, eps_s)
    # free-running fp32, no Fourier features: 1e-5 per step
    assert rel_linf(xh[-1], g["sample"]) < 1e-5 and rel_linf(mus[-1], g["mu_last"]) < 1e-5
    for a, b in [(mus[:, :4], g["mus_first4"]), (xh[:, :4], g["x_hats_first4"]), (ys[:, :4], g["ys_first4"])]:
        for i in range(a.shape[0]):
            assert rel_linf(a[i], b[i]) < 1e-5, (i, rel_linf(a[i], b[i]))
