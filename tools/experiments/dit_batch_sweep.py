import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ["WHICH"] = "none"
import importlib.util
spec = importlib.util.spec_from_file_location("sb", "tools/secondary_bench.py")
sb = importlib.util.module_from_spec(spec); spec.loader.exec_module(sb)
shape = (3, 32, 32)
bsi = sb.make_bsi(sb.dit(shape, 2), shape, 16)
g = torch.Generator(sb.dev).manual_seed(0)
with torch.no_grad():
    for rep in range(2):
        for b in (32, 64, 96, 128, 192, 256, 384, 512):
            bsi.sample(b, g); torch.cuda.synchronize()
            t0 = time.perf_counter(); bsi.sample(b, g); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"B={b}: k=16 {b/dt:.1f} img/s = {b/dt*17*161.46/1e3:.0f} TF", flush=True)
