"""Parity at the STATED tolerances and at every BASELINE configuration's own geometry.

Stated tolerances (BASELINE.json north_star), bf16-MFMA path against the
reference's fp32 results:
  * train_loss batch mean, relative    <= 1e-4
  * train_loss per sample, relative    <= 1e-3
  * teacher-forced x_hat, rel-Linf     <= 1e-2
Every test reports the error it achieved (tests.util.report)."""
import contextlib
from unittest import mock

import pytest
import torch

from oracle import bsi_oracle as bo
from oracle import dit_oracle as do
from tests.util import CALIB, CALIB_UNET, calib_unet_weights, calib_weights, golden, max_rel, rel_linf, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


@contextlib.contextmanager
def replay_noise(**queues):
    qs = {k: list(v) for k, v in queues.items()}

    def pop(name):
        def f(*a, **kw):
            return qs[name].pop(0).to(kw.get("device", DEV))
        return f

    with contextlib.ExitStack() as st:
        for name in qs:
            st.enter_context(mock.patch.object(torch, name, side_effect=pop(name)))
        yield
    assert all(len(v) == 0 for v in qs.values()), "not all recorded draws were consumed"


def make_bsi(model, shape, k=128):
    from bsi_amd import BSI, Discretization
    return BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k, preconditioning="edm",
               discretization=Discretization.image_8bit()).to(DEV)


def native_dit(W, shape, ps, dim, depth, heads, dropout=None):
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    m = DenoisingDiT(shape, ps, dim, depth, heads, dropout=dropout, fourier_features=FourierFeatures(n_min=6, n_max=8))
    m.load_state_dict(W)
    return m.to(DEV)


def grad_errors(model, ref_grads, floor_grads=None, top=6):
    """Per-tensor relative L2 error of the parameter gradients and the error of the global norm.  Returns (worst, norm_err, table):
    `table` lists EVERY tensor, worst first, as (err, name, |ref|, floor) -- assert_grad_table walks all of them; `top` only
    bounds what callers put into the report() record (table[:top]) -- where floor = rel. L2 distance of `floor_grads` (the oracle
    evaluated with bf16-rounded GEMM / convolution operands, `md=torch.bfloat16`) from the fp32 oracle for that tensor -- what
    rounding the operands alone costs, before any kernel is involved."""
    rows, sq, sqr = [], 0.0, 0.0
    for name, p in model.named_parameters():
        r = ref_grads[name].double()
        got = p.grad.detach().cpu().double()
        sq += float((got ** 2).sum())
        sqr += float((r ** 2).sum())
        err = float((got - r).norm() / r.norm().clamp_min(1e-30))
        fl = None
        if floor_grads is not None:
            fl = float((floor_grads[name].double() - r).norm() / r.norm().clamp_min(1e-30))
        rows.append((err, name, float(r.norm()), fl))
    rows.sort(reverse=True)
    return rows[0][:2], abs((sq / sqr) ** 0.5 - 1), rows


GRAD_TOL = 1e-2  # stated bound on the relative L2 error of every parameter-gradient tensor


def assert_grad_table(table, tol=GRAD_TOL):
    """Every gradient tensor within `tol` of the fp32 oracle -- or, where rounding the operands to bf16 ALONE moves the oracle's
    own gradient by more than tol / 2 (floor column), within 1.5 x that floor: the kernel may not add to what the operand
    precision the north_star prescribes (bf16 MFMA) costs by itself."""
    for err, name, _, fl in table:
        if err <= tol:
            continue
        assert fl is not None and fl > tol / 2 and err <= 1.5 * fl, (name, err, fl)


def test_calibration_point_train_loss_at_stated_tolerance():
    """SURVEY Appendix F calibration model (DiT dim 128, depth 4, heads 2, patch 4, adaLN un-zeroed, B = 64), golden g11
    generated from the reference: train_loss mean <= 1e-4, per sample <= 1e-3, teacher-forced x_hat <= 1e-2."""
    c = CALIB
    g = golden("g11_calib_dit")
    W = calib_weights()
    model = native_dit(W, c["shape"], c["patch_size"], c["dim"], c["depth"], c["heads"]).eval()
    bsi = make_bsi(model, c["shape"])
    with torch.no_grad(), replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps"]]):
        loss = bsi.train_loss(g["x"].to(DEV)).cpu()
    per = ((loss.double() - g["loss"].double()).abs() / g["loss"].double().abs())
    mean_err = abs(float(loss.double().mean()) / float(g["loss"].double().mean()) - 1)
    mean_err64 = abs(float(loss.double().mean()) / float(g["loss_fp64"].mean()) - 1)
    with torch.no_grad():
        xh = bsi._predict_x(g["tf_mu"].to(DEV), g["tf_t"].to(DEV)).cpu()
    tf = [rel_linf(xh[i], g["tf_xhat"][i]) for i in range(len(xh))]
    report("calibration_train_loss", mean_rel=mean_err, mean_rel_vs_fp64=mean_err64, per_sample_max=per.max(),
           per_sample_median=per.median(), teacher_forced_xhat_max=max(tf), teacher_forced_xhat=[float(v) for v in tf],
           stated="mean 1e-4, per-sample 1e-3, x_hat 1e-2")
    assert mean_err <= 1e-4, mean_err
    assert float(per.max()) <= 1e-3, float(per.max())
    assert max(tf) <= 1e-2, tf


def test_calibration_point_gradients():
    """Same model in train() mode (no dropout): loss through the taping forward at the same tolerance, gradients of every
    parameter against autograd through the fp32 oracle (itself pinned to the reference's per-tensor gradient norms)."""
    c = CALIB
    g = golden("g11_calib_dit")
    W = calib_weights()
    model = native_dit(W, c["shape"], c["patch_size"], c["dim"], c["depth"], c["heads"]).train()
    bsi = make_bsi(model, c["shape"])
    with replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps"]]):
        loss = bsi.train_loss(g["x"].to(DEV))
    loss.mean().backward()
    lc = loss.detach().cpu()
    per = float(max_rel(lc, g["loss"]))
    mean_err = abs(float(lc.double().mean()) / float(g["loss"].double().mean()) - 1)
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    f = lambda mu, t: do.dit_forward(Wr, mu, t, patch_size=c["patch_size"], dim=c["dim"], depth=c["depth"],  # noqa: E731
                                     heads=c["heads"], ff=c["ff"])
    ref = bo.BSIOracle(f, data_shape=c["shape"], k=128).train_loss(g["x"], g["offset"], g["perm"], g["eps"])
    ref.mean().backward()
    worst, norm_err, table = grad_errors(model, {k: v.grad for k, v in Wr.items()})
    gn = sum(float(p.grad.double().pow(2).sum()) for p in model.parameters()) ** 0.5
    norm_vs_ref = abs(gn / float(g["grad_norm"]) - 1)
    report("calibration_gradients", train_mode_mean_rel=mean_err, train_mode_per_sample_max=per,
           worst_tensor_rel_l2=worst[0], worst_tensor=worst[1], grad_norm_rel=norm_err, grad_norm_rel_vs_reference=norm_vs_ref)
    assert mean_err <= 1e-4 and per <= 1e-3, (mean_err, per)
    assert_grad_table(table)
    assert norm_vs_ref < 1e-3, norm_vs_ref


def test_full_size_dit_l2_train_loss_and_gradients_vs_oracle():
    """DiT-L/2 (config/experiment/imagenet32.yaml:33-39, the BASELINE model: dim 1024, depth 24, heads 16, patch 2) at
    B = 4: BSI.train_loss and the gradient of its mean for every one of the 478.6 M parameters against autograd through the
    fp32 CPU oracle."""
    shape, ps, dim, depth, heads, B = (3, 32, 32), 2, 1024, 24, 16, 4
    W = do.dit_random_weights(shape, ps, dim, depth, ff=(6, 8), seed=0)
    model = native_dit(W, shape, ps, dim, depth, heads).train()
    bsi = make_bsi(model, shape)
    gen = torch.Generator().manual_seed(21)
    x = (torch.randint(0, 256, (B, *shape), generator=gen).float() / 255) * 2 - 1
    off, perm, eps = torch.rand((), generator=gen), torch.randperm(B, generator=gen), torch.randn((B, *shape), generator=gen)
    with replay_noise(rand=[off], randperm=[perm], randn=[eps]):
        loss = bsi.train_loss(x.to(DEV))
    loss.mean().backward()
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    f = lambda mu, t: do.dit_forward(Wr, mu, t, patch_size=ps, dim=dim, depth=depth, heads=heads, ff=(6, 8))  # noqa: E731
    ref = bo.BSIOracle(f, data_shape=shape, k=128).train_loss(x, off, perm, eps)
    ref.mean().backward()
    lc = loss.detach().cpu()
    per = float(max_rel(lc, ref.detach()))
    mean_err = abs(float(lc.double().mean()) / float(ref.detach().double().mean()) - 1)
    worst, norm_err, table = grad_errors(model, {k: v.grad for k, v in Wr.items()})
    report("dit_l2_full_size_train", B=B, mean_rel=mean_err, per_sample_max=per, worst_tensor_rel_l2=worst[0],
           worst_tensor=worst[1], grad_norm_rel=norm_err)
    assert per <= 1e-3 and mean_err <= 1e-4, (per, mean_err)   # the stated tolerances
    assert_grad_table(table)
    assert norm_err < 1e-3, norm_err


def test_config5_dit_l4_imagenet64_geometry():
    """Config 5 at its own geometry (config/experiment/imagenet64.yaml:33-39): DiT-L/4 on 3x64x64, k = 256.  One
    preconditioned evaluation against the fp32 oracle; `elbo` on the native path against the oracle fed with the same
    draws; a k = 256 schedule sampled for its first steps teacher-free with determinism / finiteness / range checks
    (bsi/bsi.py:291-310)."""
    shape, ps, dim, depth, heads = (3, 64, 64), 4, 1024, 24, 16
    W = do.dit_random_weights(shape, ps, dim, depth, ff=(6, 8), seed=5)
    model = native_dit(W, shape, ps, dim, depth, heads).eval()
    bsi = make_bsi(model, shape, k=256)
    gen = torch.Generator().manual_seed(31)
    mu = torch.randn((2, *shape), generator=gen) * 2
    t = torch.tensor([0.15, 0.85])
    f = lambda a, b: do.dit_forward(W, a, b, patch_size=ps, dim=dim, depth=depth, heads=heads, ff=(6, 8))  # noqa: E731
    o = bo.BSIOracle(f, data_shape=shape, k=256, discretization=bo.Disc.image_8bit())
    with torch.no_grad():
        got = bsi._predict_x(mu.to(DEV), t.to(DEV)).cpu()
        ref = o.predict_x(mu, t)
    e_fwd = rel_linf(got, ref)
    # ELBO: n_recon = 1, n_measure = 2 on two images, same draws on both sides (draw order of bsi/bsi.py:152-182)
    B, nr, nm = 2, 1, 2
    x = (torch.randint(0, 256, (B, *shape), generator=gen).float() / 255) * 2 - 1
    eps_r = torch.randn((nr, B, *shape), generator=gen)
    off, perm = torch.rand((), generator=gen), torch.randperm(nm * B, generator=gen)
    eps_m = torch.randn((nm, B, *shape), generator=gen)
    with torch.no_grad(), replay_noise(randn=[eps_r, eps_m], rand=[off], randperm=[perm]):
        elbo, bpd, extra = bsi.elbo(x.to(DEV), nr, nm)
    with torch.no_grad():
        lr = o.reconstruction_loss(x, eps_r)
        lm = o.inf_measurement_loss(x, off, perm, eps_m)
        relbo, rbpd, _ = o.assemble_elbo(lr, lm)
    e_bpd = float(max_rel(bpd.cpu(), rbpd))
    e_lm = float(max_rel(extra["l_measure"].cpu(), lm))
    e_lr = float(max_rel(extra["l_recon"].cpu(), lr))
    # first 3 steps and the final prediction of the k = 256 schedule: deterministic, finite, inside the data range margin
    tt = torch.cat([bsi.default_schedule[:4], bsi.default_schedule[-1:]])
    with torch.no_grad():
        a = bsi.sample(2, torch.Generator(DEV).manual_seed(9), t=tt)
        b = bsi.sample(2, torch.Generator(DEV).manual_seed(9), t=tt)
    report("config5_dit_l4_64x64", forward_rel_linf=e_fwd, elbo_bpd_max_rel=e_bpd, l_measure_max_rel=e_lm, l_recon_max_rel=e_lr,
           schedule_len=len(bsi.default_schedule))
    assert len(bsi.default_schedule) == 257
    assert e_fwd < 2e-3, e_fwd   # x_hat of one evaluation, rel. L-inf (stated 1e-2 for trajectories)
    assert e_lm < 1e-3 and e_lr < 1e-3 and e_bpd < 1e-4, (e_lm, e_lr, e_bpd)   # per-sample loss terms 1e-3, the bpd mean 1e-4
    assert torch.equal(a, b) and torch.isfinite(a).all() and a.shape == (2, *shape)


def test_drivers_vs_reference_fixtures_on_the_native_path():
    """f1/f2 on the HIP path: schedules through `bsi_amd.BSI.p_lambda.cdf` (kernel) against the arrays the reference's
    scripts/generate_samples.py:117-152 produced; the ELBO loop's reduction on per-sample values that come from the
    native `elbo` equals the reference formula (eval_elbo.py:150-160) applied to the same values."""
    import os
    import numpy as np
    from bsi_amd import drivers as D
    from tests.util import weights
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_drivers.npz"))
    m = DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8))
    m.load_state_dict(weights("dit_ff"))
    bsi = make_bsi(m.to(DEV).eval(), (3, 16, 16), k=8)
    worst = 0.0
    for k in (8, 128):
        for name in D.SCHEDULES:
            t = D.sampling_schedule(bsi, name, k)
            assert t.device.type == "cuda"
            err = float(np.abs(t.cpu().double().numpy() - z[f"sched_{name}_{k}"].astype(np.float64)).max())
            worst = max(worst, err)
            assert err <= 5e-7, (name, k, err)   # fp32 log on the device vs torch CPU: a few ulp of t in [0, 1]
    gen = torch.Generator().manual_seed(3)
    x = (torch.randint(0, 256, (7, 3, 16, 16), generator=gen).float() / 255) * 2 - 1
    acc = D.evaluate_elbo(bsi, [x[:4], x[4:]], 2, 2, ["inf", 8], torch.Generator(DEV).manual_seed(11))
    for kk in ("inf", 8):
        a = acc[kk]
        n = len(a.bpd)
        assert n == 7 and abs(a.mean_var() - (a.bpd.var(ddof=1) + a.bpd_var.mean()) / n) < 1e-15
    report("drivers_native", schedule_max_abs_err=worst)


def native_unet(W, shape, dim, levels, dropout=0.1):
    from bsi_amd.models.pos_emb import NyquistPositionalEmbedding
    from bsi_amd.models.vdm_unet import DenoisingVDMUNet
    from bsi_amd.nn import FourierFeatures
    m = DenoisingVDMUNet(shape, NyquistPositionalEmbedding(32, 100), "silu", dim, levels, 4, n_attention_heads=1, dropout=dropout,
                         fourier_features=FourierFeatures(n_min=6, n_max=8))
    m.load_state_dict(W)
    return m.to(DEV)


def test_unet_calibration_point_at_stated_tolerance():
    """Config 2's denoiser family at the stated tolerances: VDM-UNet dim 128, 2 levels, 16x16, B = 64 (golden g13 generated
    from the reference class): train_loss mean <= 1e-4, per sample <= 1e-3, teacher-forced x_hat <= 1e-2; gradients of every
    parameter against autograd through the oracle."""
    from oracle import unet_oracle as uo
    c = CALIB_UNET
    g = golden("g13_calib_unet")
    W = calib_unet_weights()
    model = native_unet(W, c["shape"], c["dim"], c["levels"]).eval()   # eval(): dropout off, as in the golden
    bsi = make_bsi(model, c["shape"])
    with replay_noise(rand=[g["offset"]], randperm=[g["perm"]], randn=[g["eps"]]):
        loss = bsi.train_loss(g["x"].to(DEV))
    loss.mean().backward()
    lc = loss.detach().cpu()
    per = ((lc.double() - g["loss"].double()).abs() / g["loss"].double().abs())
    mean_err = abs(float(lc.double().mean()) / float(g["loss"].double().mean()) - 1)
    with torch.no_grad():
        xh = bsi._predict_x(g["tf_mu"].to(DEV), g["tf_t"].to(DEV)).cpu()
    tf = [rel_linf(xh[i], g["tf_xhat"][i]) for i in range(len(xh))]
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    f = lambda mu, t: uo.unet_forward(Wr, mu, t, levels=c["levels"], ff=c["ff"], has_dropout_slot=True)  # noqa: E731
    ref = bo.BSIOracle(f, data_shape=c["shape"], k=128).train_loss(g["x"], g["offset"], g["perm"], g["eps"])
    ref.mean().backward()
    worst, norm_err, table = grad_errors(model, {k: v.grad for k, v in Wr.items()})
    gn = sum(float(p.grad.double().pow(2).sum()) for p in model.parameters()) ** 0.5
    report("unet_calibration", mean_rel=mean_err, per_sample_max=per.max(), per_sample_median=per.median(),
           teacher_forced_xhat_max=max(tf), worst_tensor_rel_l2=worst[0], worst_tensor=worst[1],
           grad_norm_rel_vs_reference=abs(gn / float(g["grad_norm"]) - 1), stated="mean 1e-4, per-sample 1e-3, x_hat 1e-2")
    assert mean_err <= 1e-4, mean_err
    assert float(per.max()) <= 1e-3, float(per.max())
    assert max(tf) <= 1e-2, tf
    assert_grad_table(table)
    assert abs(gn / float(g["grad_norm"]) - 1) < 1e-3


def test_full_size_unet_train_loss_and_gradients_vs_oracle():
    """VDM-UNet of config/experiment/cifar10-vdm.yaml:32-39 (dim 128, 32 levels, 1 head, 3x32x32, 28.1 M parameters) at B = 2 in
    eval() mode: BSI.train_loss and the gradient of its mean for every parameter against autograd through the fp32 oracle."""
    from oracle import unet_oracle as uo
    shape, dim, levels, B = (3, 32, 32), 128, 32, 2
    W = uo.unet_random_weights(shape, dim, levels, seed=0, ff=(6, 8))
    model = native_unet(W, shape, dim, levels).eval()
    bsi = make_bsi(model, shape)
    gen = torch.Generator().manual_seed(41)
    x = (torch.randint(0, 256, (B, *shape), generator=gen).float() / 255) * 2 - 1
    off, perm, eps = torch.rand((), generator=gen), torch.randperm(B, generator=gen), torch.randn((B, *shape), generator=gen)
    with replay_noise(rand=[off], randperm=[perm], randn=[eps]):
        loss = bsi.train_loss(x.to(DEV))
    loss.mean().backward()
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    Wr = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    f = lambda mu, t: uo.unet_forward(Wr, mu, t, levels=levels, ff=(6, 8), has_dropout_slot=True)  # noqa: E731
    ref = bo.BSIOracle(f, data_shape=shape, k=128).train_loss(x, off, perm, eps)
    ref.mean().backward()
    lc = loss.detach().cpu()
    per = float(max_rel(lc, ref.detach()))
    mean_err = abs(float(lc.double().mean()) / float(ref.detach().double().mean()) - 1)
    # what rounding the convolution / GEMM operands to bf16 costs by itself: the SAME oracle with md = bf16 (operands rounded,
    # fp32 accumulation, everything else fp32) against the fp32 oracle, per gradient tensor
    Wb = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    fb = lambda mu, t: uo.unet_forward(Wb, mu, t, levels=levels, ff=(6, 8), has_dropout_slot=True, md=torch.bfloat16)  # noqa: E731
    bo.BSIOracle(fb, data_shape=shape, k=128).train_loss(x, off, perm, eps).mean().backward()
    worst, norm_err, table = grad_errors(model, {k: v.grad for k, v in Wr.items()}, {k: v.grad for k, v in Wb.items()}, top=8)
    report("unet_full_size_train", B=B, mean_rel=mean_err, per_sample_max=per, worst_tensor_rel_l2=worst[0],
           worst_tensor=worst[1], grad_norm_rel=norm_err,
           worst_tensors=[{"name": n, "rel_l2": e, "ref_norm": rn, "bf16_operand_floor": fl} for e, n, rn, fl in table[:8]],
           tensors_checked=len(table), tensors_over_tol=sum(1 for r in table if r[0] > GRAD_TOL))
    assert per <= 1e-3 and mean_err <= 1e-4, (per, mean_err)   # the stated tolerances
    assert_grad_table(table)
    assert norm_err < 1e-3, norm_err
