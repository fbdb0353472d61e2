"""Which stage of the DiT evaluation depends on the batch size?  adaLN table and engine forward for 256 images against their halves."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tests")
from tests.test_hip_fullsize_properties import _dit, _unet, DEV
B = 256
shape = (3, 32, 32)
for make in (_dit, _unet):
    m = make()
    gen = torch.Generator(DEV).manual_seed(11)
    mu = torch.randn((B, *shape), device=DEV, generator=gen) * 2
    t = torch.rand(B, device=DEV, generator=gen)
    with torch.no_grad():
        mod = m.adaln_table(t)
        modh = torch.cat([m.adaln_table(t[:128]), m.adaln_table(t[128:])])
        print(make.__name__, "table 256 vs halves: equal", torch.equal(mod, modh), float((mod - modh).abs().max()))
        full = m.forward_native(mu, mod)
        halves = torch.cat([m.forward_native(mu[:128].contiguous(), mod[:128].contiguous()), m.forward_native(mu[128:].contiguous(), mod[128:].contiguous())])
        print(make.__name__, "forward with the SAME table: equal", torch.equal(full, halves), float((full - halves).abs().max()))
        one = m.forward_native(mu, mod[:1].contiguous())
        oneh = torch.cat([m.forward_native(mu[:128].contiguous(), mod[:1].contiguous()), m.forward_native(mu[128:].contiguous(), mod[:1].contiguous())])
        print(make.__name__, "forward with ONE table row: equal", torch.equal(one, oneh), float((one - oneh).abs().max()))
        q = torch.cat([m.forward_native(mu[i:i + 64].contiguous(), mod[i:i + 64].contiguous()) for i in range(0, B, 64)])
        print(make.__name__, "forward quarters: equal", torch.equal(full, q), float((full - q).abs().max()))
