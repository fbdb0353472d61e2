"""Plain-PyTorch yardstick for the secondary configuration (CIFAR10 VDM-UNet: dim 128, 32 levels, no down-sampling, 1 attention head):
nn.Conv2d / nn.GroupNorm / F.scaled_dot_product_attention, bf16 autocast, channels_last, eager or torch.compile (COMPILE=1).
  sampling : 129 evaluations of 512 images -> images/s      training : fwd + bwd + clip + fused AdamW + EMA at batch 128 -> steps/s"""
import os, sys, time, torch
import torch.nn as nn
import torch.nn.functional as F

dev = torch.device("cuda")
dim, levels, cdim = 128, 32, 128


class Res(nn.Module):
    def __init__(self, din, dout, p=0.1):
        super().__init__()
        self.n1, self.c1 = nn.GroupNorm(32, din), nn.Conv2d(din, dout, 3, padding=1)
        self.film = nn.Linear(cdim, 2 * dout)
        self.c2 = nn.Conv2d(dout, dout, 3, padding=1)
        self.skip = nn.Conv2d(din, dout, 1) if din != dout else nn.Identity()
        self.p = p

    def forward(self, x, c):
        h = self.c1(F.silu(self.n1(x)))
        sc, sh = self.film(c)[:, :, None, None].chunk(2, dim=1)
        h = F.dropout(F.silu(h * (1 + sc) + sh), self.p, self.training)
        return self.skip(x) + self.c2(h)


class Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.n, self.qkv, self.out = nn.GroupNorm(32, dim), nn.Conv2d(dim, 3 * dim, 3, padding=1), nn.Conv2d(dim, dim, 3, padding=1)

    def forward(self, x):
        B, C, H, W = x.shape
        q, k, v = self.qkv(self.n(x)).reshape(B, 3, 1, C, H * W).permute(1, 0, 2, 4, 3)
        a = F.scaled_dot_product_attention(q, k, v)
        return x + self.out(a.permute(0, 1, 3, 2).reshape(B, C, H, W))


class UNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.cmap = nn.Sequential(nn.Linear(32, cdim), nn.SiLU(), nn.Linear(cdim, cdim), nn.SiLU())
        self.enc, self.dec = nn.Conv2d(21, dim, 3, padding=1), nn.Conv2d(dim, 3, 1)
        self.down = nn.ModuleList(Res(dim, dim) for _ in range(levels))
        self.mid1, self.attn, self.mid2 = Res(dim, dim), Attn(), Res(dim, dim)
        self.up = nn.ModuleList(Res(2 * dim, dim) for _ in range(levels))

    def forward(self, x, temb):
        c = self.cmap(temb)
        h = self.enc(x)
        skips = []
        for b in self.down:
            h = b(h, c)
            skips.append(h)
        h = self.mid2(self.attn(self.mid1(h, c)), c)
        for b in self.up:
            h = b(torch.cat([h, skips.pop()], dim=1), c)
        return self.dec(h)


torch.manual_seed(0)
m = UNet().to(dev).to(memory_format=torch.channels_last)
MODE = "eager"
if os.environ.get("COMPILE"):
    m = torch.compile(m)
    MODE = "torch.compile"
B = 512
x = torch.randn((B, 21, 32, 32), device=dev).to(memory_format=torch.channels_last)
te = torch.randn((B, 32), device=dev)
m.eval()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3): m(x, te)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): m(x, te)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"torch {MODE}, bf16 autocast, channels_last, VDM-UNet forward at {B} images: {1e3 * dt:.1f} ms per evaluation -> {B / (129 * dt):.1f} images/s at 129 evaluations per image")
B = 128
x, te, tgt = x[:B].contiguous(memory_format=torch.channels_last), te[:B], torch.randn((B, 3, 32, 32), device=dev)
m.train()
opt = torch.optim.AdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.99), weight_decay=1e-2, fused=True)
ema = [p.detach().clone() for p in m.parameters()]
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.mse_loss(m(x, te).float(), tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    torch._foreach_lerp_(ema, [p.detach() for p in m.parameters()], 1e-4)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"torch {MODE}, bf16 autocast, channels_last, VDM-UNet train step at batch {B}: {1e3 * dt:.1f} ms -> {1 / dt:.2f} steps/s")
