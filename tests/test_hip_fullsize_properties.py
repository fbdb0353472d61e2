"""Size-independent properties of the hot path AT the benchmark's workload (BASELINE configs 2-4: DiT-L/2 and the CIFAR-10
VDM-UNet, 256 images per call): the oracle cannot run these sizes in test time, but the persistent kernels' tile walks, XCD
partitions, band orders and per-image statistics only show their bugs there.  Every image of a batch is an independent
chain in the reference (bsi/bsi.py:312-336; nothing in bsi/models/dit.py or bsi/models/vdm_unet.py mixes batch entries),
so on identical inputs

  * a batch of 256 must equal the concatenation of its two halves evaluated alone      (batch independence),
  * a permuted batch must give the permuted result                                     (permutation equivariance),
  * a repeated call must reproduce itself                                               (determinism),

BIT FOR BIT: all three only re-order work between workgroups, never the arithmetic of one output element.  Checked for one
preconditioned denoiser evaluation with per-image times and for a 4-step sampling chain with injected noise."""
import pytest
import torch

from tests.util import report

pytestmark = pytest.mark.gpu
DEV = "cuda"
B = 256


def _bsi(model, shape, k=128):
    from bsi_amd import BSI, Discretization
    return BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k, preconditioning="edm",
               discretization=Discretization.image_8bit()).to(DEV)


def _dit():
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    torch.manual_seed(0)
    m = DenoisingDiT((3, 32, 32), 2, 1024, 24, 16, dropout=0.05, fourier_features=FourierFeatures(n_min=6, n_max=8))
    with torch.no_grad():
        for blk in m.dit.blocks:  # the reference zero-initialises this layer (dit.py:84-85): un-zero it, or blocks are the identity
            blk.adaLN_modulation[-1].weight.normal_(0, 0.02)
    return m.to(DEV).eval()


def _unet():
    from bsi_amd.models.pos_emb import NyquistPositionalEmbedding
    from bsi_amd.models.vdm_unet import DenoisingVDMUNet
    from bsi_amd.nn import FourierFeatures
    torch.manual_seed(0)
    return DenoisingVDMUNet((3, 32, 32), NyquistPositionalEmbedding(32, 100), "silu", 128, 32, 4, n_attention_heads=1, dropout=0.1,
                            fourier_features=FourierFeatures(n_min=6, n_max=8)).to(DEV).eval()


@pytest.mark.parametrize("make", [_dit, _unet], ids=["dit_l2", "vdm_unet"])
def test_full_batch_evaluation_properties(make):
    shape = (3, 32, 32)
    model = make()
    bsi = _bsi(model, shape)
    gen = torch.Generator(DEV).manual_seed(11)
    mu = torch.randn((B, *shape), device=DEV, generator=gen) * 2
    t = torch.rand(B, device=DEV, generator=gen)
    h, cut = B // 2, 171
    with torch.no_grad():
        # the denoiser engine on a GIVEN conditioning table (adaLN rows / FiLM coefficients, one row per image): bit for bit
        mod = model.adaln_table(t)
        f = lambda lo, hi, m=mod: model.forward_native(mu[lo:hi].contiguous(), m[lo:hi].contiguous())  # noqa: E731
        full = f(0, B)
        assert bool(torch.isfinite(full).all())
        assert torch.equal(full, f(0, B)), "not deterministic"
        assert torch.equal(full, torch.cat([f(0, h), f(h, B)])), "batch differs from its halves"
        assert torch.equal(full, torch.cat([f(0, cut), f(cut, B)])), "batch differs from an uneven split (tail tiles)"
        perm = torch.randperm(B, device=DEV, generator=gen)
        assert torch.equal(model.forward_native(mu[perm].contiguous(), mod[perm].contiguous()), full[perm]), "not permutation equivariant"
        # the whole preconditioned evaluation with per-image times (train_loss / elbo path, bsi.py:375-388).  The conditioning
        # table comes from GEMMs whose M is the number of rows; the small-M kernel and the ring kernels sum in the same order
        # (accumulators start at the bias in both), so this path, too, is bit for bit independent of the batch -- on both sides of
        # the kernels' M = 128 threshold (171 + 85)
        x = bsi._predict_x(mu, t)
        assert torch.equal(x, bsi._predict_x(mu, t))
        assert torch.equal(model.adaln_table(t), torch.cat([model.adaln_table(t[:cut]), model.adaln_table(t[cut:])])), "conditioning table"
        assert torch.equal(x, torch.cat([bsi._predict_x(mu[:h], t[:h]), bsi._predict_x(mu[h:], t[h:])]))
        assert torch.equal(x, torch.cat([bsi._predict_x(mu[:cut], t[:cut]), bsi._predict_x(mu[cut:], t[cut:])]))
        report("fullsize_batch_independence", model=make.__name__.strip("_"), batch=B, engine_bit_exact=True, per_image_time_path_bit_exact=True)


@pytest.mark.parametrize("make", [_dit, _unet], ids=["dit_l2", "vdm_unet"])
def test_full_batch_sampling_chain_properties(make):
    shape = (3, 32, 32)
    k = 4
    bsi = _bsi(make(), shape, k=k)
    gen = torch.Generator(DEV).manual_seed(12)
    eps0 = torch.randn((B, *shape), device=DEV, generator=gen)
    eps = torch.randn((k, B, *shape), device=DEV, generator=gen)
    t = torch.linspace(0, 1, k + 1, device=DEV)
    with torch.no_grad():
        full = bsi._run_chain(B, None, t, history=False, noise=(eps0, eps))
        assert bool(torch.isfinite(full).all()) and float(full.abs().max()) < 50
        h = B // 2
        a = bsi._run_chain(h, None, t, history=False, noise=(eps0[:h].contiguous(), eps[:, :h].contiguous()))
        b = bsi._run_chain(h, None, t, history=False, noise=(eps0[h:].contiguous(), eps[:, h:].contiguous()))
        assert torch.equal(full, torch.cat([a, b])), "sampling 256 images differs from sampling its halves"
        assert torch.equal(full, bsi._run_chain(B, None, t, history=False, noise=(eps0, eps))), "chain not deterministic"
        # the captured-graph path replays the same chain
        if make is _dit:
            g1 = torch.Generator(DEV).manual_seed(3)
            g2 = torch.Generator(DEV).manual_seed(3)
            assert torch.equal(bsi.sample(B, g1, t=t), bsi.sample(B, g2, t=t, graph=True))


@pytest.mark.parametrize("make,batch", [(_dit, 64), (_unet, 128)], ids=["dit_l2", "vdm_unet"])
def test_full_size_gradient_of_a_batch_is_the_mean_of_its_shards(make, batch):
    """The premise of the data-parallel train step (bsi/tasks/bsi.py:187-194 under DDP): with the same per-sample
    lambda and noise, the gradient of the mean loss over a batch equals the average of the gradients over its two equal shards
    (what two ranks would all-reduce) -- at the full model size, dropout off.  Per-sample losses agree closely (the conditioning
    table depends on the row count, see above); gradients to the accuracy of bf16 operands summed in a different split order."""
    from unittest import mock
    shape = (3, 32, 32)
    model = make()  # eval mode: the differentiable path runs with dropout off (masks are drawn per call)
    bsi = _bsi(model, shape)
    gen = torch.Generator(DEV).manual_seed(21)
    x = (torch.randint(0, 256, (batch, *shape), device=DEV, generator=gen).float() / 255) * 2 - 1
    lam = torch.exp(torch.rand(batch, device=DEV, generator=gen) * 18.4 - 4.6).clamp(1e-2, 1e6 - 1)
    eps = torch.randn((batch, *shape), device=DEV, generator=gen)

    def run(lo, hi):
        for p in model.parameters():
            p.grad = None
        real_randn = torch.randn
        with mock.patch.object(bsi, "_sample_lambda", lambda n, b, g=None: lam[lo:hi].reshape(1, -1).clone()), \
                mock.patch.object(torch, "randn", lambda *a, **kw: eps[lo:hi].clone() if a and tuple(a[0]) == (hi - lo, *shape) else real_randn(*a, **kw)):
            per = bsi.train_loss(x[lo:hi])
        per.mean().backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).double()
        return per.detach().double(), g

    h = batch // 2
    per_full, g_full = run(0, batch)
    per_a, g_a = run(0, h)
    per_b, g_b = run(h, batch)
    assert bool(torch.isfinite(g_full).all()) and float(g_full.norm()) > 0
    per_sh = torch.cat([per_a, per_b])
    per_rel = float(((per_full - per_sh).abs() / per_full.abs().clamp_min(1e-6)).max())
    g_sh = 0.5 * (g_a + g_b)
    rel = float((g_full - g_sh).norm() / g_full.norm())
    report("fullsize_shard_equivalence", model=make.__name__.strip("_"), batch=batch, per_sample_max_rel=per_rel, flat_gradient_rel_l2=rel)
    assert per_rel < 1e-3, per_rel  # measured: 0 (both shard sizes take the same conditioning GEMM kernel)
    assert rel < 1e-4, rel          # measured: 3e-6 (DiT-L/2), 7e-7 (UNet)


@pytest.mark.parametrize("make,batch", [(_dit, 32), (_unet, 64)], ids=["dit_l2", "vdm_unet"])
def test_train_steps_are_bit_reproducible(make, batch):
    """Two runs of three optimizer steps (forward + backward + clip + AdamW + EMA, dropout ON) from the same initial state, data and
    generator seed end in bit-identical parameters: the backward has no atomics -- modulation / FiLM gradients go to per-slab
    planes summed in fixed order, GroupNorm / final-LayerNorm / decoder gradients to per-image or per-block rows summed by
    reduce_slabs, weight gradients to split-M slabs -- and the dropout masks are counter based."""
    from bsi_amd.dp import DPTrainer
    shape = (3, 32, 32)
    finals = []
    for _ in range(2):
        torch.manual_seed(0)
        model = make().train()
        trainer = DPTrainer(_bsi(model, shape), lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0)
        gen = torch.Generator(DEV).manual_seed(5)
        x = (torch.randint(0, 256, (batch, *shape), device=DEV, generator=torch.Generator(DEV).manual_seed(1)).float() / 255) * 2 - 1
        losses = [float(trainer.train_step(x, gen)) for _ in range(3)]
        torch.cuda.synchronize()
        finals.append((losses, trainer.fp.flat.clone(), trainer.ema_fp.flat.clone() if trainer.ema_fp is not None else None))
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1]), float((finals[0][1] - finals[1][1]).abs().max())
    if finals[0][2] is not None:
        assert torch.equal(finals[0][2], finals[1][2])
    report("train_step_reproducibility", model=make.__name__.strip("_"), batch=batch, steps=3, bit_identical=True)


def test_train_step_is_bit_identical_with_the_tile_queue_and_a_cu_reserve():
    """The data-parallel step's tile queue (persistent GEMM: tile tickets instead of static shares, on all CUs) changes which workgroup
    computes a tile, not a bit of it: three optimizer steps of the full-size DiT-L/2 at 32 images (fc1 / qkv / fc2 input gradient: more
    tiles than CUs) end in the same parameters and EMA with the queue on as with static shares.  (The CU reserve moves the split counts
    of the weight-gradient GEMM -- another summation order -- so it is compared loosely, not bit for bit.)"""
    from bsi_amd.dp import DPTrainer
    shape = (3, 32, 32)
    finals = {}
    for name, kw in (("static", dict(tile_queue=False)), ("queue", dict(tile_queue=True)), ("queue + reserve", dict(tile_queue=True, cu_reserve=16))):
        torch.manual_seed(0)
        model = _dit().train()
        # rehearse=(1, 0): the layout of a single rank, with the switches of an exchanging step in force during the backward
        trainer = DPTrainer(_bsi(model, shape), lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0, rehearse=(1, 0), **kw)
        assert trainer.tile_queue == kw["tile_queue"] and trainer.cu_reserve == kw.get("cu_reserve", 0)
        gen = torch.Generator(DEV).manual_seed(5)
        x = (torch.randint(0, 256, (32, *shape), device=DEV, generator=torch.Generator(DEV).manual_seed(1)).float() / 255) * 2 - 1
        losses = [float(trainer.train_step(x, gen)) for _ in range(3)]
        torch.cuda.synchronize()
        finals[name] = (losses, trainer.fp.flat.clone(), trainer.ema_fp.flat.clone())
        del trainer, model
        torch.cuda.empty_cache()
    a, b, c = finals["static"], finals["queue"], finals["queue + reserve"]
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    rel = float((c[1] - a[1]).abs().max() / a[1].abs().max())
    assert rel < 2e-3, rel  # measured 2.8e-4: Adam's first steps move a weight by ~lr whatever the size of its gradient, so last-bit differences of tiny gradients show
    report("train_step_tile_queue", model="dit", batch=32, steps=3, queue_bit_identical=True, reserve_rel_linf=rel)


def test_train_step_is_bit_identical_under_both_attention_backward_schedules():
    """Three optimizer steps of the full-size DiT-L/2 at 32 images (512 (image, head) pairs: two per compute unit, dropout on, the qkv
    bias sums riding on the attention backward) end in the same parameters and EMA with the backward's wave groups half a trip apart
    (the default) as in lock step: the skew moves work between barriers, not a bit of the result."""
    from bsi_amd import _native as N
    from bsi_amd.dp import DPTrainer
    shape = (3, 32, 32)
    finals = {}
    prev = N.lib().bsi_set_attention_bwd_skew(1)
    try:
        for on in (1, 0):
            N.lib().bsi_set_attention_bwd_skew(on)
            torch.manual_seed(0)
            model = _dit().train()
            trainer = DPTrainer(_bsi(model, shape), lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0)
            gen = torch.Generator(DEV).manual_seed(5)
            x = (torch.randint(0, 256, (32, *shape), device=DEV, generator=torch.Generator(DEV).manual_seed(1)).float() / 255) * 2 - 1
            losses = [float(trainer.train_step(x, gen)) for _ in range(3)]
            torch.cuda.synchronize()
            finals[on] = (losses, trainer.fp.flat.clone(), trainer.ema_fp.flat.clone())
            del trainer, model
            torch.cuda.empty_cache()
    finally:
        N.lib().bsi_set_attention_bwd_skew(prev)
    a, b = finals[1], finals[0]
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    report("train_step_attention_bwd_schedules", model="dit", batch=32, steps=3, bit_identical=True)
