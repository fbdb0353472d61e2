import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests.util import golden, weights
from tests.test_hip_dit import replay_noise
from tests.test_hip_dit import make_bsi, DEV
from bsi_amd.models.dit import DenoisingDiT
from bsi_amd.nn import FourierFeatures
p, B, d, heads, depth, side = 0.3, 4, 128, 2, 2, 32
shape = (3, side, side)
W = weights("dit_ff")
model = DenoisingDiT(shape, 2, d, depth, heads, dropout=p, fourier_features=FourierFeatures(n_min=6, n_max=8))
model.load_state_dict(W); model = model.to(DEV).train()
bsi = make_bsi(model, shape)
gen = torch.Generator().manual_seed(77)
x = (torch.round(255 * torch.rand((B, *shape), generator=gen)) / 255) * 2 - 1
off, perm, eps = torch.rand((), generator=gen), torch.randperm(B, generator=gen), torch.randn((B, *shape), generator=gen)
torch.manual_seed(123)
with replay_noise(rand=[off], randperm=[perm], randn=[eps]):
    loss = bsi.train_loss(x.to(DEV))
print("loss", loss)
loss.mean().backward()
for n, q in model.named_parameters():
    g = q.grad
    print(n, "nan" if not torch.isfinite(g).all() else "ok", float(g.float().norm()))

# ---- where does the first NaN appear?  (tape layout mirrors carve_tape / block_tape of dit_train.hip)
def au(v, a=256): return (v + a - 1) // a * a
M, dim, kpad, tokens = B * 256, d, 128, 256
off = au(M * kpad * 2) + au(M * dim * 4) + au(B * depth * 6 * dim * 4) + au(B * dim * 2) + au(depth * B * dim * 4) + au(depth * B * dim * 2)
names = [("xn1", M * dim * 2, torch.bfloat16), ("qkv", M * 3 * dim * 2, torch.bfloat16), ("ao", M * dim * 2, torch.bfloat16), ("d1", M * dim * 2, torch.bfloat16),
         ("xn2", M * dim * 2, torch.bfloat16), ("hp", M * 4 * dim * 2, torch.bfloat16), ("h", M * 4 * dim * 2, torch.bfloat16), ("d2", M * dim * 2, torch.bfloat16),
         ("lse", B * heads * tokens * 4, torch.float32), ("xa", M * dim * 4, torch.float32), ("xb", M * dim * 4, torch.float32), ("sa", M * 8, torch.float32),
         ("sb", M * 8, torch.float32), ("maskw", B * heads * 8192, torch.uint8)]
block_bytes = sum(au(n) for _, n, _ in names)
import torch as _t
_orig_empty = _t.empty
stash = {}
def _empty(*a, **k):
    r = _orig_empty(*a, **k)
    if k.get("dtype") == _t.uint8 and r.numel() > (1 << 20): stash.setdefault("tapes", []).append(r)
    return r
_t.empty = _empty
model.zero_grad()
torch.manual_seed(123)
with replay_noise(rand=[off_ := off * 0 + torch.rand((), generator=gen)], randperm=[perm], randn=[eps]):
    loss = bsi.train_loss(x.to(DEV))
_t.empty = _orig_empty
tape = max(stash["tapes"], key=lambda t: t.numel())
print("tape bytes", tape.numel(), "expected", off + block_bytes * depth)
for l in range(depth):
    o = off + l * block_bytes
    for nm, nbytes, dt in names:
        t_ = tape[o:o + nbytes].view(dt)
        if dt != torch.uint8:
            print(l, nm, "finite" if bool(torch.isfinite(t_.float()).all()) else "NAN", float(t_.float().abs().max()))
        else:
            print(l, nm, "popcount fraction", float(((t_.view(torch.int64).unsqueeze(-1) >> torch.arange(64, device=t_.device)) & 1).float().mean()))
        o += au(nbytes)

# ---- replay block 0's attention on the engine's own qkv through the internal entry point
import ctypes as C
from bsi_amd import _native as N
lib = N.lib()
class DropCfg(C.Structure):
    _fields_ = [("thr", C.c_uint), ("s0", C.c_uint), ("s1", C.c_uint), ("scale", C.c_float)]
fwd = getattr(lib, "_Z23bsi_attention_fwd_trainPKviiiiiPviPf7DropCfgS1_S1_b")
fwd.restype = C.c_int
fwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, DropCfg, C.c_void_p, C.c_void_p, C.c_bool]
def mix32(x):
    x &= 0xffffffff; x ^= x >> 16; x = (x * 0x7feb352d) & 0xffffffff; x ^= x >> 15; x = (x * 0x846ca68b) & 0xffffffff; x ^= x >> 16; return x
def make_drop(p, seed, site):
    s = (seed + 0x9E3779B97F4A7C15 * (site + 1)) & 0xffffffffffffffff
    t = p * 4294967296.0 + 32768.0
    thr = max(4294967295 if t >= 4294967295.0 else int(t), 65536)
    return DropCfg(thr, mix32(s & 0xffffffff), mix32(((s >> 32) ^ 0x85ebca6b) & 0xffffffff), 65536.0 / (65536 - (thr >> 16)))
o0 = off
qkv = tape[o0 + au(M * dim * 2): o0 + au(M * dim * 2) + M * 3 * dim * 2].view(torch.bfloat16).clone()
seed = (torch.initial_seed() * 0x9E3779B1 + model._drop_calls * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
for sd in (seed, 1234567):
    dc = make_drop(p, sd, 0)
    for words in (True, False):
        out = torch.zeros((M, dim), device="cuda", dtype=torch.bfloat16); lse = torch.zeros(B * heads * 256, device="cuda")
        mw = torch.zeros(B * heads * 8192, dtype=torch.uint8, device="cuda")
        N.check(fwd(N.ptr(qkv), 3 * dim, B, 256, heads, 64, N.ptr(out), dim, N.ptr(lse), dc, N.stream(), N.ptr(mw) if words else None, False))
        torch.cuda.synchronize()
        print("seed", sd, "words", words, "out finite", bool(torch.isfinite(out.float()).all()), "lse finite", bool(torch.isfinite(lse).all()), "dc", dc.thr, dc.s0, dc.s1, dc.scale)
