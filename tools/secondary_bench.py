#!/usr/bin/env python
"""Secondary measurements of SURVEY §8(d) on one GPU (one JSON object per line on stdout):
  * DiT-L/2 32x32 k=128 sampling at 64 / 256 / 512 images per call (the headline bench.py line uses 128),
  * DiT-L/4 64x64 k=256 sampling (config/experiment/imagenet64.yaml:33-39),
  * ELBO / bpd evaluation throughput (elbo with 1 reconstruction + 1 measurement sample = 2 forwards per image),
  * VDM-UNet k=128 sampling (config/experiment/cifar10-vdm.yaml:32-39),
  * VDM-UNet train steps/s at global batch 128 (fwd + bwd + clip + AdamW + EMA, dropout 0.1) on one GPU.
FLOP figures are SURVEY §8's probe values (2*MAC)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsi_amd import BSI, Discretization  # noqa: E402
from bsi_amd.models.dit import DenoisingDiT  # noqa: E402
from bsi_amd.models.pos_emb import NyquistPositionalEmbedding  # noqa: E402
from bsi_amd.models.vdm_unet import DenoisingVDMUNet  # noqa: E402
from bsi_amd.nn import FourierFeatures  # noqa: E402

dev = torch.device("cuda", 0)
WHICH = set(os.environ.get("WHICH", "dit32,dit64,elbo,unet,unet_train").split(","))


def make_bsi(model, shape, k):
    return BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k, preconditioning="edm",
               discretization=Discretization.image_8bit()).to(dev)


def dit(shape, patch):
    torch.manual_seed(0)
    m = DenoisingDiT(shape, patch, 1024, 24, 16, dropout=0.05, fourier_features=FourierFeatures(n_min=6, n_max=8))
    with torch.no_grad():
        for blk in m.dit.blocks:  # un-zero the adaLN output layer so that blocks are not the identity
            blk.adaLN_modulation[-1].weight.normal_(0, 0.02)
    return m.to(dev).eval()


def timed(fn, reps=1):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def emit(**kw):
    print(json.dumps(kw), flush=True)


with torch.no_grad():
    if "dit32" in WHICH or "elbo" in WHICH:
        shape = (3, 32, 32)
        bsi = make_bsi(dit(shape, 2), shape, 128)
        g = torch.Generator(dev).manual_seed(0)
        if "dit32" in WHICH:
            for b in (64, 256, 512):
                dt, out = timed(lambda: bsi.sample(b, g))
                assert torch.isfinite(out).all()
                emit(what="DiT-L/2 32x32 BSI.sample k=128", images_per_call=b, images_per_s=b / dt,
                     model_tflops=b / dt * 129 * 161.46 / 1e3)
        if "elbo" in WHICH:
            b = 512
            x = (torch.randint(0, 256, (b, *shape), device=dev).float() / 255) * 2 - 1
            dt, (elbo, bpd, _) = timed(lambda: bsi.elbo(x, 1, 1, g), reps=3)
            assert torch.isfinite(bpd).all()
            emit(what="DiT-L/2 32x32 BSI.elbo(n_recon=1, n_measure=1)", images_per_call=b, images_per_s=b / dt,
                 model_tflops=b / dt * 2 * 161.46 / 1e3, bpd_mean=float(bpd.mean()))
        del bsi
    if "dit64" in WHICH:
        shape = (3, 64, 64)
        bsi = make_bsi(dit(shape, 4), shape, int(os.environ.get("DIT64_K", "256")))
        g = torch.Generator(dev).manual_seed(0)
        b = 128
        dt, out = timed(lambda: bsi.sample(b, g))
        assert torch.isfinite(out).all()
        emit(what="DiT-L/4 64x64 BSI.sample k=256", images_per_call=b, images_per_s=b / dt,
             model_tflops=b / dt * 257 * 161.61 / 1e3)
        del bsi
    if "unet" in WHICH:
        shape = (3, 32, 32)
        torch.manual_seed(0)
        m = DenoisingVDMUNet(shape, NyquistPositionalEmbedding(32, 100), "silu", 128, 32, 4, n_attention_heads=1,
                             dropout=0.1, fourier_features=FourierFeatures(n_min=6, n_max=8)).to(dev).eval()
        bsi = make_bsi(m, shape, 128)
        g = torch.Generator(dev).manual_seed(0)
        b = 256
        dt, out = timed(lambda: bsi.sample(b, g))
        assert torch.isfinite(out).all()
        emit(what="VDM-UNet(dim128, levels32) 32x32 BSI.sample k=128", images_per_call=b, images_per_s=b / dt,
             model_tflops=b / dt * 129 * 53.47 / 1e3)

if "unet_train" in WHICH:
    from bsi_amd.dp import DPTrainer
    shape = (3, 32, 32)
    torch.manual_seed(0)
    m = DenoisingVDMUNet(shape, NyquistPositionalEmbedding(32, 100), "silu", 128, 32, 4, n_attention_heads=1, dropout=0.1,
                         fourier_features=FourierFeatures(n_min=6, n_max=8)).to(dev).train()
    bsi = make_bsi(m, shape, 128)
    tr = DPTrainer(bsi, lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0)
    g = torch.Generator(dev).manual_seed(0)
    b = int(os.environ.get("UNET_TRAIN_BATCH", "128"))
    x = (torch.randint(0, 256, (b, *shape), device=dev).float() / 255) * 2 - 1
    dt, loss = timed(lambda: tr.train_step(x, g), reps=3)
    assert torch.isfinite(loss)
    emit(what="VDM-UNet train step (fwd+bwd+clip+AdamW+EMA, dropout 0.1)", global_batch=b, steps_per_s=1 / dt,
         ms_per_step=1e3 * dt, images_per_s=b / dt, model_tflops=b / dt * 3 * 53.47 / 1e3, loss=float(loss))
