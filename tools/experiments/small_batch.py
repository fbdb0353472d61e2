import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
from bsi_amd import BSI, Discretization
model, shape = bench.build_model(dev)
bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=128, preconditioning="edm", discretization=Discretization.image_8bit()).to(dev)
g = torch.Generator(dev).manual_seed(0)
for B in (1, 4, 16, 64):
    with torch.no_grad():
        bsi.sample(B, g); torch.cuda.synchronize()
        t0 = time.perf_counter(); bsi.sample(B, g); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B}: {dt*1e3:.0f} ms per sample() call, {dt/129*1e3:.2f} ms per denoiser step, {B/dt:.1f} images/s", flush=True)
