#!/usr/bin/env python
"""Secondary benchmark: images/s of BSI.sample (k=128) with the VDM-UNet of config/experiment/cifar10-vdm.yaml
(dim 128, levels 32, 1 attention head) on one GPU; 53.47 GFLOP per evaluation per image."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsi_amd import BSI, Discretization  # noqa: E402
from bsi_amd.models.pos_emb import NyquistPositionalEmbedding  # noqa: E402
from bsi_amd.models.vdm_unet import DenoisingVDMUNet  # noqa: E402
from bsi_amd.nn import FourierFeatures  # noqa: E402

B = int(os.environ.get("B", "256"))
K = int(os.environ.get("K", "128"))
dev = torch.device("cuda", 0)
shape = (3, 32, 32)
torch.manual_seed(0)
m = DenoisingVDMUNet(shape, NyquistPositionalEmbedding(32, 100), "silu", 128, 32, 4, n_attention_heads=1, dropout=0.1,
                     fourier_features=FourierFeatures(n_min=6, n_max=8)).to(dev).eval()
bsi = BSI(m, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=K, preconditioning="edm",
          discretization=Discretization.image_8bit()).to(dev)
g = torch.Generator(dev).manual_seed(0)
with torch.no_grad():
    bsi.sample(B, g, t=torch.linspace(0, 1, 5, device=dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = bsi.sample(B, g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
assert torch.isfinite(out).all()
print(f"UNet BSI.sample k={K} B={B}: {B / dt:.2f} images/s, {B / dt * (K + 1) * 53.47 / 1e3:.0f} model TFLOP/s")
