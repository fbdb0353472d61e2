#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE (imported through tools/ref_shim.py)
on CPU with seeded generators.  Runs only in the build container (needs /root/reference);
the resulting ``tests/golden/*.npz`` are plain arrays (inputs, weights under the reference's
state-dict keys, expected outputs) and are committed.  Nothing of the reference's source is
stored.  Re-run:  python tools/gen_golden.py

Golden sets follow SURVEY.md Appendix B (G1..G8).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

ref = ref_shim.load_all()
META = {"torch": torch.__version__, "generator": "tools/gen_golden.py"}


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrs):
    arrs = {k: npy(v) for k, v in arrs.items()}
    arrs["_meta_torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


def sd(model, prefix="W."):
    return {prefix + k: v for k, v in model.state_dict().items()}


_SAVED_W = {}


def save_weights(tag, model):
    """Weights of the shared small models are stored once (tests/golden/w_<tag>.npz)."""
    cur = {k: npy(v) for k, v in model.state_dict().items()}
    if tag in _SAVED_W:
        assert all(np.array_equal(cur[k], _SAVED_W[tag][k]) for k in cur), tag
        return
    _SAVED_W[tag] = cur
    save("w_" + tag, **cur)


class TinyConv(torch.nn.Module):
    """The README denoiser (README.md:21-28): Conv2d(4->3, 3x3) on cat(mu, t-plane)."""

    def __init__(self):
        super().__init__()
        self.layer = torch.nn.Conv2d(in_channels=4, out_channels=3, kernel_size=3, padding=1)

    def forward(self, mu, t):
        t = torch.movedim(t.expand((1, *mu.shape[-2:], len(t))), -1, 0)
        return self.layer(torch.cat((mu, t), dim=-3))


def make_bsi(model, shape, k=16, dtype=torch.float32, disc=True):
    b = ref.BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k,
                preconditioning="edm",
                discretization=ref.Discretization.image_8bit() if disc else None)
    return b.to(dtype)


def data(B, shape, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((B, *shape), generator=g)
    return (torch.round(255 * u) / 255) * 2 - 1


def unzero_adaln(model, seed=7):
    g = torch.Generator().manual_seed(seed)
    for blk in model.dit.blocks:
        lin = blk.adaLN_modulation[-1]
        with torch.no_grad():
            lin.weight.copy_(0.02 * torch.randn(lin.weight.shape, generator=g))
            lin.bias.copy_(0.02 * torch.randn(lin.bias.shape, generator=g))


def small_dit(shape=(3, 16, 16), ff=True, seed=1, dim=128, depth=2, heads=2, patch=2):
    torch.manual_seed(seed)
    m = ref.dit.DenoisingDiT(shape, patch, dim, depth, heads, dropout=None,
                             fourier_features=ref.nn.FourierFeatures(n_min=6, n_max=8) if ff else None)
    unzero_adaln(m)
    return m.eval()


def small_unet(shape=(3, 8, 8), ff=True, seed=2, dim=64, levels=1, dropout=0.1):
    torch.manual_seed(seed)
    m = ref.vdm_unet.DenoisingVDMUNet(
        shape, ref.pos_emb.NyquistPositionalEmbedding(32, 100), "silu", dim, levels, 4,
        n_attention_heads=1, dropout=dropout, downsampling_attention=False,
        fourier_features=ref.nn.FourierFeatures(n_min=6, n_max=8) if ff else None)
    # perturb the norm affine parameters so they are exercised
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.add_(0.05 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.add_(0.05 * torch.randn(mod.bias.shape, generator=g))
    return m.eval()


# ---------------------------------------------------------------------------------------
def g1_tables():
    b = make_bsi(TinyConv(), (3, 8, 8))
    t = torch.cat([torch.linspace(0, 1, 33), torch.tensor([1e-6, 0.5 + 1e-4, 1 - 1e-6])])
    lam = b.p_lambda.icdf(t)
    c_skip, c_out, c_in = b._edm_preconditioning(t)
    b64 = make_bsi(TinyConv(), (3, 8, 8), dtype=torch.float64)
    t64 = t.double()
    lam64 = b64.p_lambda.icdf(t64)
    cs64, co64, ci64 = b64._edm_preconditioning(t64)
    save("g1_tables", t=t, lam=lam, cdf_lam=b.p_lambda.cdf(lam), rpdf=b.p_lambda.reciprocal_pdf(lam),
         c_skip=c_skip, c_out=c_out, c_in=c_in,
         ln_low=np.float64(b.p_lambda.ln_low), delta=np.float64(b.p_lambda.diff_ln_high_ln_low),
         lam64=lam64, c_skip64=cs64, c_out64=co64, c_in64=ci64,
         ln_low64=np.float64(b64.p_lambda.ln_low), delta64=np.float64(b64.p_lambda.diff_ln_high_ln_low),
         default_schedule=b.default_schedule)


def g2_g3_lambda_and_q():
    b = make_bsi(TinyConv(), (3, 8, 8))
    out = {}
    for n, B in [(1, 8), (3, 5)]:
        g = torch.Generator().manual_seed(11 + n)
        lam = b._sample_lambda(n, B, g)
        g = torch.Generator().manual_seed(11 + n)
        off = torch.rand((), generator=g)
        perm = torch.randperm(n * B, generator=g)
        out[f"lam_{n}_{B}"] = lam
        out[f"offset_{n}_{B}"] = off
        out[f"perm_{n}_{B}"] = perm
    x = data(5, (3, 8, 8), 3)
    lam = out["lam_3_5"]
    g = torch.Generator().manual_seed(5)
    mu = b._sample_q_mu_lambda(x, lam, g)
    g = torch.Generator().manual_seed(5)
    eps = torch.randn((3, 5, 3, 8, 8), generator=g)
    save("g2g3_lambda_q", x=x, q_lam=lam, q_eps=eps, q_mu=mu, **out)


def train_loss_case(name, model, shape, B, seed, wtag=None):
    b = make_bsi(model, shape)
    x = data(B, shape, seed)
    g = torch.Generator().manual_seed(seed + 1)
    loss = b.train_loss(x, g)
    model.zero_grad()
    loss.mean().backward()
    grads = {"G." + k: p.grad for k, p in model.named_parameters()}
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))
    g = torch.Generator().manual_seed(seed + 1)
    off = torch.rand((), generator=g)
    perm = torch.randperm(B, generator=g)
    eps = torch.randn((B, *shape), generator=g)
    # fp64 run of the same thing (reference in double) for tolerance calibration
    m64 = type(model).__new__(type(model))
    import copy
    m64 = copy.deepcopy(model).double()
    b64 = make_bsi(m64, shape, dtype=torch.float64)
    lam64 = b64.p_lambda.icdf(torch.remainder(perm.double() / (1 + B) + off.double(), 1))
    mu64 = torch.addcmul(((lam64 - b64.lambda_0) / lam64).view(-1, 1, 1, 1) * x.double(),
                         torch.rsqrt(lam64).view(-1, 1, 1, 1), eps.double())
    xh64 = b64._predict_x(mu64, b64.p_lambda.cdf(lam64))
    loss64 = b64.p_lambda.reciprocal_pdf(lam64) * (x.double() - xh64).square().flatten(1).mean(1)
    save(name, x=x, offset=off, perm=perm, eps=eps, loss=loss, loss_mean=loss.mean(), grad_norm=gn,
         loss_fp64=loss64, **(sd(model) if wtag is None else {}), **grads)
    if wtag is not None:
        save_weights(wtag, model)


def g4_train_loss():
    torch.manual_seed(0)
    train_loss_case("g4_train_tinyconv", TinyConv(), (3, 8, 8), 8, 20)
    train_loss_case("g4_train_dit", small_dit(), (3, 16, 16), 4, 30, "dit_ff")
    train_loss_case("g4_train_dit_noff", small_dit(ff=False, seed=5), (3, 16, 16), 4, 35, "dit_noff")
    train_loss_case("g4_train_unet", small_unet(), (3, 8, 8), 4, 40, "unet_ff")


def history_case(name, model, shape, n, k, seed, wtag=None):
    b = make_bsi(model, shape, k=k)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        mus, x_hats, ys = b.sample_history(n, g)
    g = torch.Generator().manual_seed(seed)
    eps0 = torch.randn((n, *shape), generator=g)
    eps = torch.stack([torch.randn((n, *shape), generator=g) for _ in range(k)])
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        smp = b.sample(n, g)
    assert torch.equal(smp, x_hats[-1])
    save(name, eps0=eps0, eps=eps, mus=mus, x_hats=x_hats, ys=ys, k=np.int64(k),
         **(sd(model) if wtag is None else {}))
    if wtag is not None:
        save_weights(wtag, model)


def g5_history():
    torch.manual_seed(3)
    history_case("g5_hist_tinyconv", TinyConv(), (3, 8, 8), 4, 16, 50)
    history_case("g5_hist_dit_noff", small_dit(ff=False, seed=5), (3, 16, 16), 2, 16, 51, "dit_noff")
    history_case("g5_hist_dit_ff", small_dit(), (3, 16, 16), 2, 16, 52, "dit_ff")
    history_case("g5_hist_unet_ff", small_unet(), (3, 8, 8), 2, 16, 53, "unet_ff")
    history_case("g5_hist_unet_noff", small_unet(ff=False, seed=6), (3, 8, 8), 2, 16, 54, "unet_noff")


def fingerprint(t):
    d = t.double()
    return torch.stack((d.sum(), d.abs().sum(), d.flatten()[0], d.flatten()[-1], (d * d).sum()))


def g14_config1():
    """BASELINE.json configs[0] at its stated size: the README denoiser on 3x32x32, batch 32, train_loss + sample k = 16
    (README.md:21-35).  The Gaussian noise (2 + 16 draws of [32, 3, 32, 32]) is NOT stored: the tests re-draw it from the seeds
    below with the same generator calls and check the stored fingerprints (torch CPU generator, same image on the GPU box)."""
    torch.manual_seed(14)
    model = TinyConv()
    shape, B, k, seed_t, seed_s = (3, 32, 32), 32, 16, 140, 141
    b = make_bsi(model, shape, k=k)
    x = data(B, shape, 142)
    g = torch.Generator().manual_seed(seed_t)
    loss = b.train_loss(x, g)
    model.zero_grad()
    loss.mean().backward()
    grads = {"G." + kk: p.grad.clone() for kk, p in model.named_parameters()}
    g = torch.Generator().manual_seed(seed_t)
    off = torch.rand((), generator=g)
    perm = torch.randperm(B, generator=g)
    eps = torch.randn((B, *shape), generator=g)
    g = torch.Generator().manual_seed(seed_s)
    with torch.no_grad():
        mus, x_hats, ys = b.sample_history(B, g)
    g = torch.Generator().manual_seed(seed_s)
    with torch.no_grad():
        smp = b.sample(B, g)
    assert torch.equal(smp, x_hats[-1])
    g = torch.Generator().manual_seed(seed_s)
    eps0 = torch.randn((B, *shape), generator=g)
    eps_s = torch.stack([torch.randn((B, *shape), generator=g) for _ in range(k)])
    save("g14_config1", x=x, seed_train=np.int64(seed_t), seed_sample=np.int64(seed_s), k=np.int64(k),
         offset=off, perm=perm, eps_fp=fingerprint(eps), eps0_fp=fingerprint(eps0), eps_steps_fp=fingerprint(eps_s),
         loss=loss.detach(), loss_mean=loss.detach().mean(), sample=smp, mu_last=mus[-1],
         x_hats_first4=x_hats[:, :4], mus_first4=mus[:, :4], ys_first4=ys[:, :4], **sd(model), **grads)


def g15_branches():
    """The two branches of the BSI surface no other set exercises (bsi/bsi.py:379-380 and 441-445).

    A. preconditioning=None: `_predict_x` is the bare denoiser.  Small DiT without Fourier features (weights w_dit_noff):
       train_loss + gradients, free-running sample_history k = 4, elbo with estimate_var.
    B. low_discrepancy_sampling=False: `_sample_lambda` draws `rand((batch, n))` and returns that TRANSPOSED (batch, n) shape
       (SURVEY Appendix D.1).  Recorded: the draw and the result for (n, B) = (3, 5); `train_loss` of the small DiT at B = 4, where
       `_sample_lambda(1, B)[0]` is ONE lambda (shape (1,)) that broadcasts over the batch together with ONE noise image
       (bsi.py:309-310,405-420); and `inf_measurement_loss` at n = B = 4 (README denoiser), the one shape for which the reference's
       broadcasting of the transposed grid against the batch is defined."""
    import copy
    shape = (3, 16, 16)
    model = small_dit(ff=False, seed=5)
    save_weights("dit_noff", model)
    out = {}
    # ---- A
    b = ref.BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=4, preconditioning=None,
                discretization=ref.Discretization.image_8bit())
    B = 4
    x = data(B, shape, 150)
    g = torch.Generator().manual_seed(151)
    loss = b.train_loss(x, g)
    model.zero_grad()
    loss.mean().backward()
    grads = {"A_G." + k: p.grad.clone() for k, p in model.named_parameters()}
    g = torch.Generator().manual_seed(151)
    out.update(A_x=x, A_offset=torch.rand((), generator=g), A_perm=torch.randperm(B, generator=g),
               A_eps=torch.randn((B, *shape), generator=g), A_loss=loss.detach(), A_loss_mean=loss.detach().mean())
    g = torch.Generator().manual_seed(152)
    with torch.no_grad():
        mus, x_hats, ys = b.sample_history(2, g)
    g = torch.Generator().manual_seed(152)
    out.update(A_eps0=torch.randn((2, *shape), generator=g), A_eps_steps=torch.stack([torch.randn((2, *shape), generator=g) for _ in range(4)]),
               A_mus=mus, A_x_hats=x_hats, A_ys=ys)
    nr, nm = 2, 3
    with torch.no_grad():
        g = torch.Generator().manual_seed(153)
        elbo, bpd, extra = b.elbo(x, nr, nm, g, estimate_var=True)
        g = torch.Generator().manual_seed(153)
        out.update(A_eps_r=torch.randn((nr, B, *shape), generator=g), A_e_offset=torch.rand((), generator=g),
                   A_e_perm=torch.randperm(nm * B, generator=g), A_eps_m=torch.randn((nm, B, *shape), generator=g),
                   A_elbo=elbo, A_bpd=bpd, A_l_recon=extra["l_recon"], A_l_measure=extra["l_measure"], A_bpd_var=extra["bpd_var"])
    # ---- B
    bp = ref.BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=4, preconditioning="edm",
                 low_discrepancy_sampling=False, discretization=ref.Discretization.image_8bit())
    g = torch.Generator().manual_seed(154)
    lam = bp._sample_lambda(3, 5, g)
    g = torch.Generator().manual_seed(154)
    u = torch.rand((5, 3), generator=g)
    assert lam.shape == (5, 3)
    out.update(B_u=u, B_lam=lam)
    g = torch.Generator().manual_seed(155)
    with torch.no_grad():
        loss_b = bp.train_loss(x, g)
    g = torch.Generator().manual_seed(155)
    u1 = torch.rand((B, 1), generator=g)
    eps1 = torch.randn((1, *shape), generator=g)
    assert loss_b.shape == (B,)
    out.update(B_t_u=u1, B_t_eps=eps1, B_t_loss=loss_b)
    tc = TinyConv()
    torch.manual_seed(156)
    tc = TinyConv()
    bt = ref.BSI(tc, data_shape=(3, 8, 8), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=4, preconditioning="edm",
                 low_discrepancy_sampling=False, discretization=ref.Discretization.image_8bit())
    xt = data(4, (3, 8, 8), 157)
    g = torch.Generator().manual_seed(158)
    with torch.no_grad():
        lm = bt.inf_measurement_loss(xt, 4, g)
    g = torch.Generator().manual_seed(158)
    um = torch.rand((4, 4), generator=g)
    epsm = torch.randn((4, 4, 3, 8, 8), generator=g)
    out.update(B_m_x=xt, B_m_u=um, B_m_eps=epsm, B_m_loss=lm, **sd(tc, "B_m_W."))
    save("g15_branches", **out, **grads)


def g6_elbo():
    torch.manual_seed(4)
    model = TinyConv()
    shape = (3, 8, 8)
    B, nr, nm, k = 5, 3, 4, 16
    b = make_bsi(model, shape, k=k)
    x = data(B, shape, 60)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -1.0, 1.0])  # edge bins
    x[1] = 1.0
    x[2] = -1.0
    with torch.no_grad():
        g = torch.Generator().manual_seed(61)
        elbo, bpd, extra = b.elbo(x, nr, nm, g, estimate_var=True)
        g = torch.Generator().manual_seed(61)
        eps_r = torch.randn((nr, B, *shape), generator=g)
        off = torch.rand((), generator=g)
        perm = torch.randperm(nm * B, generator=g)
        eps_m = torch.randn((nm, B, *shape), generator=g)

        g = torch.Generator().manual_seed(62)
        felbo, fbpd, fextra = b.finite_elbo(x, nr, nm, g, t=torch.linspace(0, 1, k + 1), estimate_var=True)
        g = torch.Generator().manual_seed(62)
        feps_r = torch.randn((nr, B, *shape), generator=g)
        fidx = torch.randint(0, k, (nm, B), generator=g)
        feps_m = torch.randn((nm, B, *shape), generator=g)

        # continuous (no discretization) reconstruction
        bc = make_bsi(model, shape, k=k, disc=False)
        g = torch.Generator().manual_seed(63)
        lrc = bc.reconstruction_loss(x, nr, g)
        g = torch.Generator().manual_seed(63)
        ceps_r = torch.randn((nr, B, *shape), generator=g)
    save("g6_elbo", x=x, eps_r=eps_r, offset=off, perm=perm, eps_m=eps_m, elbo=elbo, bpd=bpd,
         l_recon=extra["l_recon"], l_measure=extra["l_measure"], bpd_var=extra["bpd_var"],
         feps_r=feps_r, fidx=fidx, feps_m=feps_m, felbo=felbo, fbpd=fbpd,
         fl_recon=fextra["l_recon"], fl_measure=fextra["l_measure"], fbpd_var=fextra["bpd_var"],
         ceps_r=ceps_r, cl_recon=lrc, **sd(model))


def g7_components():
    out = {}
    # positional embeddings
    for size, rate in [(1024, 1000), (32, 100), (512, 32), (64, 16)]:
        pe = ref.pos_emb.NyquistPositionalEmbedding(size, rate)
        t = torch.cat([torch.linspace(0, 1, 9), torch.tensor([0.123456, 0.999])])
        out[f"pe_{size}_{rate}_t"] = t
        out[f"pe_{size}_{rate}_scale"] = pe.scale
        out[f"pe_{size}_{rate}_bias"] = pe.bias
        out[f"pe_{size}_{rate}_out"] = pe(t)
    # Fourier features
    ffm = ref.nn.FourierFeatures(n_min=6, n_max=8)
    g = torch.Generator().manual_seed(70)
    x = torch.randn((2, 3, 4, 4), generator=g) * 3
    out["ff_x"] = x
    out["ff_out"] = ffm(x, dim=1)
    out["ff_out64"] = ffm.double()(x.double(), dim=1)
    # DiT block and attention
    torch.manual_seed(71)
    blk = ref.dit.DiTBlock(128, 2, mlp_ratio=4, dropout=None).eval()
    with torch.no_grad():
        blk.adaLN_modulation[-1].weight.normal_(0, 0.02)
        blk.adaLN_modulation[-1].bias.normal_(0, 0.02)
    xt = torch.randn((2, 64, 128), generator=g)
    c = torch.randn((2, 128), generator=g)
    with torch.no_grad():
        out["blk_x"], out["blk_c"], out["blk_out"] = xt, c, blk(xt, c)
        out["attn_out"] = blk.attn(xt)
    out.update(sd(blk, "BLK."))
    # patch pos embedding of a DiT
    d = small_dit()
    out["dit16_pos"] = d.dit.patch_pos_embedding
    # Residual blocks
    from functools import partial
    Norm = partial(torch.nn.GroupNorm, 32)
    for din, tag in [(64, "rb64"), (128, "rb128")]:
        torch.manual_seed(72)
        rb = ref.nn.ResidualBlock(din, 64, c_dim=128, ActFn=torch.nn.SiLU, Norm=Norm, dropout=0.1,
                                  attention=False).eval()
        xi = torch.randn((2, din, 8, 8), generator=g)
        ci = torch.randn((2, 128), generator=g)
        with torch.no_grad():
            out[f"{tag}_x"], out[f"{tag}_c"], out[f"{tag}_out"] = xi, ci, rb(xi, ci)
        out.update(sd(rb, f"{tag.upper()}."))
    torch.manual_seed(73)
    at = ref.nn.Attention2D(64, heads=1).eval()
    xa = torch.randn((2, 64, 8, 8), generator=g)
    with torch.no_grad():
        out["a2d_x"], out["a2d_out"] = xa, at(xa)
    out.update(sd(at, "A2D."))
    save("g7_components", **out)

    # full model forwards
    for name, m, shape, wtag in [("g7_dit_fwd", small_dit(), (3, 16, 16), "dit_ff"),
                                 ("g7_unet_fwd", small_unet(), (3, 8, 8), "unet_ff")]:
        mu = torch.randn((3, *shape), generator=g) * 2
        t = torch.tensor([0.0, 0.37, 1.0])
        with torch.no_grad():
            y = m(mu, t)
            y64 = m.double()(mu.double(), t.double())
            m.float()
        save(name, mu=mu, t=t, out=y, out64=y64)
        save_weights(wtag, m)


def g8_optimizer():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_ema", os.path.join(ref_shim.REF_ROOT, "bsi", "tasks", "ema_pytorch.py"))
    ema_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ema_mod)
    torch.manual_seed(80)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    ema = ema_mod.EMA(model, beta=0.9999, update_after_step=1000, update_every=1,
                      include_online_model=False, use_foreach=True)
    decays = []
    steps = []
    for _ in range(2100):
        s_before = int(ema.step)
        ema.update()
        # decay actually used at this update: get_current_decay() after the increment
        steps.append(s_before)
        decays.append(float(ema.get_current_decay()))
    # AdamW + clip on a small problem, 3 steps; EMA lerp with a fixed decay
    torch.manual_seed(81)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2)
    g = torch.Generator().manual_seed(82)
    rec = {}
    for i, (k_, p) in enumerate(model.named_parameters()):
        rec[f"p0.{k_}"] = p.detach().clone()
    for step in range(1, 4):
        xin = torch.randn((16, 6), generator=g) * 10
        loss = (model(xin) ** 2).sum()
        opt.zero_grad()
        loss.backward()
        for k_, p in model.named_parameters():
            rec[f"g{step}.{k_}"] = p.grad.detach().clone()
        tn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        rec[f"norm{step}"] = tn
        opt.step()
        for k_, p in model.named_parameters():
            rec[f"p{step}.{k_}"] = p.detach().clone()
    # lerp semantics of the EMA update: ema.lerp_(model, 1 - decay)
    a = torch.randn((7,), generator=g)
    bq = torch.randn((7,), generator=g)
    rec["lerp_tgt"], rec["lerp_src"] = a.clone(), bq
    a.lerp_(bq, 1.0 - 0.9)
    rec["lerp_out_0.9"] = a
    # LR schedule of the reference (bsi/lr_scheduler.py:35-58) stepped once per optimizer step
    spec = importlib.util.spec_from_file_location("ref_lr", os.path.join(ref_shim.REF_ROOT, "bsi", "lr_scheduler.py"))
    lr_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr_mod)
    prm = torch.nn.Parameter(torch.zeros(1))
    o2 = torch.optim.AdamW([prm], lr=5e-4)
    sch = lr_mod.WarmUpCosineAnnealing(o2, warmup_steps=10, max_steps=60, start_lr=1e-8, end_lr=5e-5)
    lrs = []
    for _ in range(60):
        lrs.append(o2.param_groups[0]["lr"])
        o2.step()
        sch.step()
    save("g8_optimizer", ema_steps=np.array(steps), ema_decays=np.array(decays), lr_schedule=np.array(lrs), **rec)


def g11_calibration():
    """SURVEY Appendix F calibration point: DiT dim 128, depth 4, heads 2, patch 4 on 3x32x32, adaLN un-zeroed, B = 64.
    The weights are NOT stored: they are `oracle.dit_oracle.dit_random_weights(..., seed=11)` (a seeded torch-CPU
    stream), loaded into the reference class here; the fixture holds a fingerprint of them so that a drifting RNG is
    noticed.  Expected values: the reference's fp32 `train_loss` (per sample), the same in fp64, the gradient norm
    per parameter tensor, and teacher-forced `_predict_x` at 8 times (fp32 and fp64)."""
    import copy
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import dit_oracle as do
    shape, ps, dim, depth, heads, B = (3, 32, 32), 4, 128, 4, 2, 64
    W = do.dit_random_weights(shape, ps, dim, depth, ff=(6, 8), seed=11)
    model = ref.dit.DenoisingDiT(shape, ps, dim, depth, heads, dropout=None,
                                 fourier_features=ref.nn.FourierFeatures(n_min=6, n_max=8))
    model.load_state_dict(W)
    model.eval()
    b = make_bsi(model, shape, k=128)
    x = data(B, shape, 110)
    g = torch.Generator().manual_seed(111)
    loss = b.train_loss(x, g)
    model.zero_grad()
    loss.mean().backward()
    gnorms = {"GN." + k: p.grad.double().norm() for k, p in model.named_parameters()}
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))
    g = torch.Generator().manual_seed(111)
    off = torch.rand((), generator=g)
    perm = torch.randperm(B, generator=g)
    eps = torch.randn((B, *shape), generator=g)
    m64 = copy.deepcopy(model).double()
    b64 = make_bsi(m64, shape, k=128, dtype=torch.float64)
    lam64 = b64.p_lambda.icdf(torch.remainder(perm.double() / (1 + B) + off.double(), 1))
    mu64 = torch.addcmul(((lam64 - b64.lambda_0) / lam64).view(-1, 1, 1, 1) * x.double(),
                         torch.rsqrt(lam64).view(-1, 1, 1, 1), eps.double())
    with torch.no_grad():
        xh64 = b64._predict_x(mu64, b64.p_lambda.cdf(lam64))
        loss64 = b64.p_lambda.reciprocal_pdf(lam64) * (x.double() - xh64).square().flatten(1).mean(1)
        # teacher-forced one-step predictions
        gt = torch.Generator().manual_seed(112)
        tt = torch.tensor([0.0, 0.05, 0.2, 0.4, 0.6, 0.8, 0.95, 1.0])
        lam_t = b.p_lambda.icdf(tt)
        xs = data(len(tt), shape, 113)
        mu_t = torch.addcmul(((lam_t - b.lambda_0) / lam_t).view(-1, 1, 1, 1) * xs, torch.rsqrt(lam_t).view(-1, 1, 1, 1),
                             torch.randn((len(tt), *shape), generator=gt))
        xh_t = b._predict_x(mu_t, tt)
        xh_t64 = b64._predict_x(mu_t.double(), tt.double())
    fp = torch.stack([torch.stack((v.double().sum(), v.double().abs().sum(), v.flatten()[0].double(),
                                   v.flatten()[-1].double())) for _, v in sorted(W.items())])
    save("g11_calib_dit", x=x, offset=off, perm=perm, eps=eps, loss=loss, loss_mean=loss.mean(), loss_fp64=loss64,
         grad_norm=gn, weight_fingerprint=fp, tf_t=tt, tf_mu=mu_t, tf_xhat=xh_t, tf_xhat64=xh_t64, **gnorms)


def g13_calibration_unet():
    """The UNet counterpart of g11: VDM-UNet dim 128, 2 levels, 1 head on 3x16x16, dropout slot present but eval(), B = 64.
    Weights = `oracle.unet_oracle.unet_random_weights(..., seed=13)` loaded into the reference class (fingerprint stored);
    expected: the reference's fp32 `train_loss`, the same in fp64, per-tensor gradient norms, teacher-forced `_predict_x`."""
    import copy
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import unet_oracle as uo
    shape, dim, levels, B = (3, 16, 16), 128, 2, 64
    W = uo.unet_random_weights(shape, dim, levels, seed=13, ff=(6, 8))
    model = ref.vdm_unet.DenoisingVDMUNet(
        shape, ref.pos_emb.NyquistPositionalEmbedding(32, 100), "silu", dim, levels, 4, n_attention_heads=1, dropout=0.1,
        downsampling_attention=False, fourier_features=ref.nn.FourierFeatures(n_min=6, n_max=8))
    model.load_state_dict(W)
    model.eval()
    b = make_bsi(model, shape, k=128)
    x = data(B, shape, 130)
    g = torch.Generator().manual_seed(131)
    loss = b.train_loss(x, g)
    model.zero_grad()
    loss.mean().backward()
    gnorms = {"GN." + k: p.grad.double().norm() for k, p in model.named_parameters()}
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))
    g = torch.Generator().manual_seed(131)
    off = torch.rand((), generator=g)
    perm = torch.randperm(B, generator=g)
    eps = torch.randn((B, *shape), generator=g)
    m64 = copy.deepcopy(model).double()
    b64 = make_bsi(m64, shape, k=128, dtype=torch.float64)
    lam64 = b64.p_lambda.icdf(torch.remainder(perm.double() / (1 + B) + off.double(), 1))
    mu64 = torch.addcmul(((lam64 - b64.lambda_0) / lam64).view(-1, 1, 1, 1) * x.double(),
                         torch.rsqrt(lam64).view(-1, 1, 1, 1), eps.double())
    with torch.no_grad():
        xh64 = b64._predict_x(mu64, b64.p_lambda.cdf(lam64))
        loss64 = b64.p_lambda.reciprocal_pdf(lam64) * (x.double() - xh64).square().flatten(1).mean(1)
        gt = torch.Generator().manual_seed(132)
        tt = torch.tensor([0.0, 0.05, 0.2, 0.4, 0.6, 0.8, 0.95, 1.0])
        lam_t = b.p_lambda.icdf(tt)
        xs = data(len(tt), shape, 133)
        mu_t = torch.addcmul(((lam_t - b.lambda_0) / lam_t).view(-1, 1, 1, 1) * xs, torch.rsqrt(lam_t).view(-1, 1, 1, 1),
                             torch.randn((len(tt), *shape), generator=gt))
        xh_t = b._predict_x(mu_t, tt)
        xh_t64 = b64._predict_x(mu_t.double(), tt.double())
    fp = torch.stack([torch.stack((v.double().sum(), v.double().abs().sum(), v.flatten()[0].double(),
                                   v.flatten()[-1].double())) for _, v in sorted(W.items())])
    save("g13_calib_unet", x=x, offset=off, perm=perm, eps=eps, loss=loss, loss_mean=loss.mean(), loss_fp64=loss64,
         grad_norm=gn, weight_fingerprint=fp, tf_t=tt, tf_mu=mu_t, tf_xhat=xh_t, tf_xhat64=xh_t64, **gnorms)


def kat_reference_tests():
    """Inputs/expected values of the reference's own four known-answer tests
    (tests/test_bsi.py:7-34, tests/models/components/test_fourier_features.py:9-28) evaluated
    by the reference in double precision (its tests/conftest.py:3-4 sets double)."""
    D = ref.Discretization
    d1 = D(0.0, 1.0, k=256)
    x1 = torch.tensor([-0.1, 0.0, 1.0, 1.0 - 1 / 256], dtype=torch.float64)
    d2 = D(-1.0, 1.0, k=5)
    b2 = d2.bin_boundaries(torch.device("cpu"), torch.float64)
    d3 = D(-1.0, 1.0, k=3)
    torch.set_default_dtype(torch.double)  # as the reference's tests/conftest.py:3-4
    ffm = ref.nn.FourierFeatures(n_min=5, n_max=6)
    torch.set_default_dtype(torch.float32)
    xf = torch.tensor([1.333, -np.e / 7], dtype=torch.float64)[None, :, None].repeat(2, 1, 3)
    save("kat_reference_tests", x1=x1, idx1=d1.bucketize(x1), b2=b2, idx2a=d2.bucketize(b2)[:-1],
         idx2b=d2.bucketize(b2 - 1e-8)[1:], b3=d3.bin_boundaries(torch.device("cpu"), torch.float32),
         ff_x=xf, ff_y=ffm(xf, dim=1),
         img8=D.image_8bit().to_8bit_image(torch.tensor([-1.2, -1.0, -0.5, 0.0, 0.999, 1.0, 1.5])))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:  # regenerate only the named sets, e.g. `python tools/gen_golden.py g11_calibration`
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    g1_tables()
    g2_g3_lambda_and_q()
    g4_train_loss()
    g5_history()
    g6_elbo()
    g7_components()
    g8_optimizer()
    g11_calibration()
    g13_calibration_unet()
    g14_config1()
    g15_branches()
    kat_reference_tests()
