"""Small-batch sampling latency: eager launches vs the whole chain replayed as one HIP graph (BSI.sample(graph=True))."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
from bsi_amd import BSI, Discretization
model, shape = bench.build_model(dev)
K = int(os.environ.get("K", "128"))
bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=K, preconditioning="edm", discretization=Discretization.image_8bit()).to(dev)
for B in [int(b) for b in os.environ.get("BS", "1,4,16").split(",")]:
    with torch.no_grad():
        g = torch.Generator(dev).manual_seed(7)
        bsi.sample(B, g); torch.cuda.synchronize()
        g = torch.Generator(dev).manual_seed(7)
        t0 = time.perf_counter(); a = bsi.sample(B, g); torch.cuda.synchronize(); te = time.perf_counter() - t0
        g = torch.Generator(dev).manual_seed(7)
        t0 = time.perf_counter(); bsi.sample(B, g, graph=True); torch.cuda.synchronize(); tc = time.perf_counter() - t0
        g = torch.Generator(dev).manual_seed(7)
        t0 = time.perf_counter(); b = bsi.sample(B, g, graph=True); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    print(f"B={B} k={K}: eager {te*1e3:.0f} ms, graph {tg*1e3:.0f} ms per call (capture + first replay {tc:.1f} s); same samples: {bool(torch.equal(a, b))}, max diff {float((a-b).abs().max()):.3g}", flush=True)
