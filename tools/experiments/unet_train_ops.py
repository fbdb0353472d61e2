#!/usr/bin/env python
"""Which host-side torch ops does one VDM-UNet train step issue (small copies / fills between the HIP engine's launches)?
torch.profiler table of the aten ops of one DPTrainer.train_step, by call count, with the source lines of the copies."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bsi_amd import BSI, Discretization  # noqa: E402
from bsi_amd.dp import DPTrainer  # noqa: E402
from bsi_amd.models.pos_emb import NyquistPositionalEmbedding  # noqa: E402
from bsi_amd.models.vdm_unet import DenoisingVDMUNet  # noqa: E402
from bsi_amd.nn import FourierFeatures  # noqa: E402

dev = torch.device("cuda", 0)
shape = (3, 32, 32)
torch.manual_seed(0)
m = DenoisingVDMUNet(shape, NyquistPositionalEmbedding(32, 100), "silu", 128, 32, 4, n_attention_heads=1, dropout=0.1,
                     fourier_features=FourierFeatures(n_min=6, n_max=8)).to(dev).train()
bsi = BSI(m, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=128, preconditioning="edm",
          discretization=Discretization.image_8bit()).to(dev)
tr = DPTrainer(bsi, lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0)
g = torch.Generator(dev).manual_seed(0)
x = (torch.randint(0, 256, (128, *shape), device=dev).float() / 255) * 2 - 1
for _ in range(2):
    tr.train_step(x, g)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.train_step(x, g)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=14, max_name_column_width=40))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=12, max_name_column_width=30, max_src_column_width=110))
