"""BSI.sample (DiT-L/2, 512 images, k = 128) launch by launch against one captured HIP graph per call: images/s."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from bsi_amd import BSI, Discretization
dev = torch.device("cuda", 0)
model, shape = bench.build_model(dev)
bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=128, preconditioning="edm",
          discretization=Discretization.image_8bit()).to(dev)
B = int(os.environ.get("B", "512"))
gen = torch.Generator(dev).manual_seed(1)
for graph in (False, True, False, True):
    with torch.no_grad():
        bsi.sample(B, gen, graph=graph)  # warm-up (captures)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            bsi.sample(B, gen, graph=graph)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"graph={graph}: {B / dt:.2f} images/s ({1e3 * dt:.1f} ms per call)", flush=True)
