"""What plain PyTorch gets on the same MI355X for the two headline workloads (a yardstick, not a parity check): a DiT-L/2 written with
nn.Linear / F.scaled_dot_product_attention / F.layer_norm, bf16 autocast as the reference trains (`config/train.yaml`), eager mode.
  sampling : 129 denoiser evaluations of 512 images (the arithmetic of BSI.sample k=128 without its wrapper ops) -> images/s
  training : forward + backward + torch.optim.AdamW(fused=True) + an EMA lerp at global batch 512, dropout 0.05 -> steps/s
Random weights, synthetic inputs.  python tools/experiments/torch_yardstick.py [B=512]"""
import math, sys, time, torch
import torch.nn as nn
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda")
dim, depth, heads, ps, C, Hh = 1024, 24, 16, 2, 3, 32
T = (Hh // ps) ** 2
P = C * ps * ps
KIN = C * (1 + 2 * 3) * ps * ps  # Fourier features 6..8: 3 octaves x (sin, cos) per channel + the input


class Block(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.qkv, self.out = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.fc1, self.fc2 = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim)
        self.ada = nn.Sequential(nn.Linear(dim, dim), nn.SiLU(), nn.Linear(dim, 6 * dim))
        self.p = p

    def forward(self, x, c):
        sa, ca, ga, sm, cm, gm = self.ada(c).unsqueeze(1).chunk(6, dim=-1)
        h = F.layer_norm(x, (dim,)) * (1 + ca) + sa
        q, k, v = self.qkv(h).reshape(x.shape[0], T, 3, heads, dim // heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v, dropout_p=self.p if self.training else 0.0)
        x = torch.addcmul(x, ga, self.out(a.transpose(1, 2).reshape(x.shape[0], T, dim)))
        h = F.layer_norm(x, (dim,)) * (1 + cm) + sm
        h = self.fc2(F.dropout(F.gelu(self.fc1(h), approximate="tanh"), self.p, self.training))
        return torch.addcmul(x, gm, h)


class DiT(nn.Module):
    def __init__(self, p=0.05):
        super().__init__()
        self.enc, self.pos = nn.Linear(KIN, dim), nn.Parameter(torch.zeros(T, dim))
        self.blocks = nn.ModuleList(Block(p) for _ in range(depth))
        self.dec = nn.Linear(dim, P)
        self.tw = nn.Parameter(torch.randn(dim))

    def forward(self, tok, t):
        c = torch.sin(t[:, None] * self.tw)
        x = self.enc(tok) + self.pos
        for b in self.blocks:
            x = b(x, c)
        return self.dec(F.layer_norm(x, (dim,)))


import os
torch.manual_seed(0)
m = DiT().to(dev)
MODE = "eager"
if os.environ.get("COMPILE"):  # the reference's `compile: true` option (bsi/tasks/bsi.py:91): torch.compile of the loss / sampling functions
    m = torch.compile(m)
    MODE = "torch.compile"
tok = torch.randn((B, T, KIN), device=dev)
t = torch.rand(B, device=dev)
# ---- sampling yardstick
m.eval()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3): m(tok, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): y = m(tok, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"torch {MODE}, bf16 autocast, DiT-L/2 forward at {B} images: {1e3 * dt:.1f} ms per evaluation -> {B / (129 * dt):.1f} images/s at 129 evaluations per image")
# ---- training yardstick
m.train()
opt = torch.optim.AdamW(m.parameters(), lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2, fused=True)
ema = [p.detach().clone() for p in m.parameters()]
tgt = torch.randn((B, T, P), device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.mse_loss(m(tok, t).float(), tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    torch._foreach_lerp_(ema, [p.detach() for p in m.parameters()], 1e-4)
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"torch {MODE}, bf16 autocast, DiT-L/2 train step (fwd + bwd + clip + fused AdamW + EMA) at batch {B}: {1e3 * dt:.1f} ms -> {1 / dt:.2f} steps/s  (peak memory {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB)")
