"""Is a full-size train step bit-reproducible?  Two DPTrainer runs from the same initial state, same data, same generator seed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_hip_fullsize_properties import _dit, _unet, _bsi, DEV
from bsi_amd.dp import DPTrainer
shape = (3, 32, 32)
for make, batch in ((_dit, 64), (_unet, 128)):
    outs = []
    for rep in range(2):
        torch.manual_seed(0)
        m = make().train()
        bsi = _bsi(m, shape)
        tr = DPTrainer(bsi, lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0)
        g = torch.Generator(DEV).manual_seed(5)
        x = (torch.randint(0, 256, (batch, *shape), device=DEV, generator=torch.Generator(DEV).manual_seed(1)).float() / 255) * 2 - 1
        losses = [float(tr.train_step(x, g)) for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((losses, tr.fp.flat.clone()))
    same = torch.equal(outs[0][1], outs[1][1])
    d = float((outs[0][1] - outs[1][1]).abs().max())
    print(make.__name__, "losses", outs[0][0], outs[1][0], "params bit-identical:", same, "max diff", d, flush=True)
