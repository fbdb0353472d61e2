#!/usr/bin/env python
"""Run a few DPTrainer steps of the ImageNet32 DiT-L/2 recipe on one GPU (for rocprofv3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bsi_amd import BSI, Discretization  # noqa: E402
from bsi_amd.dp import DPTrainer  # noqa: E402

B = int(os.environ.get("B", "512"))
STEPS = int(os.environ.get("STEPS", "3"))
dev = torch.device("cuda", 0)
model, shape = bench.build_model(dev)
bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=128, preconditioning="edm",
          discretization=Discretization.image_8bit()).to(dev)
model.train()
tr = DPTrainer(bsi, lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0)
g = torch.Generator(dev).manual_seed(0)
x = (torch.round(255 * torch.rand((B, *shape), device=dev, generator=g)) / 255) * 2 - 1
tr.train_step(x, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loss = tr.train_step(x, g)
torch.cuda.synchronize()
print(f"B={B}: {1e3 * (time.perf_counter() - t0) / STEPS:.1f} ms/step, loss {float(loss):.4f}")
