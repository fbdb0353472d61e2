#!/bin/bash
# Lazy store of the DiT residual row (dit_engine.hip): checksum of one BSI.sample under both forms (must be equal) and an alternating
# timing A/B of the headline loop on one box.  BSI_DIT_EAGER_RESID=1 = one store per branch (the form of rounds 1-3).
ck='import torch, hashlib, bench
from bsi_amd import BSI, Discretization
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
model, shape = bench.build_model(dev)
bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=4, preconditioning="edm", discretization=Discretization.image_8bit()).to(dev)
g = torch.Generator(device=dev).manual_seed(7)
x = bsi.sample(64, g)
print("sample sha", hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16])'
BSI_DIT_EAGER_RESID=1 python -c "$ck" | sed 's/^/eager: /'
python -c "$ck" | sed 's/^/lazy:  /'
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); print("img/s %.2f  ms/step %.1f" % (j["value"], j["ms_per_step"]))'
for r in 1 2 3; do
  BSI_DIT_EAGER_RESID=1 python bench.py --k 8 --steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 --no-secondary 2>/dev/null | python -c "$pick" | sed 's/^/eager: /'
  python bench.py --k 8 --steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 --no-secondary 2>/dev/null | python -c "$pick" | sed 's/^/lazy:  /'
done
python bench.py --k 8 --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 --no-secondary --breakdown 2>&1 >/dev/null | grep -v amdgpu
