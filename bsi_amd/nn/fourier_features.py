"""Mirror of bsi/nn/fourier_features.py:5-36 on the native kernel `bsi_fourier_features`."""
import torch
from torch import Tensor, nn

from .. import _native as N


class FourierFeatures(nn.Module):
    """Fourier features of the VDM paper: sin(2 pi 2^n x + {0, pi/2}), n = n_min..n_max.

    Same constructor (`n_min`, `n_max`, swallowed `**kwargs`), same non-persistent buffers
    (`coefs`, `offsets`) and same channel order (input channel, n, offset) as the reference.
    """

    def __init__(self, *, n_min: int, n_max: int, **kwargs):
        super().__init__()
        self.n_min = n_min
        self.n_max = n_max
        ns = torch.arange(n_min, n_max + 1)
        self.register_buffer("coefs", 2 * torch.pi * 2**ns, persistent=False)
        self.register_buffer("offsets", torch.tensor([0, torch.pi / 2]), persistent=False)

    def n_features(self):
        return len(self.coefs) * len(self.offsets)

    def forward(self, x: Tensor, *, dim: int) -> Tensor:
        assert dim >= 0, "Implementation expects a non-negative dimension index"
        if x.dtype != torch.float32:
            raise RuntimeError("bsi_amd.FourierFeatures: the native kernel computes in fp32")
        x = x.contiguous()
        shape = list(x.shape)
        outer = 1
        for s in shape[:dim]:
            outer *= s
        inner = 1
        for s in shape[dim + 1:]:
            inner *= s
        C = shape[dim]
        out_shape = shape[:dim] + [C * self.n_features()] + shape[dim + 1:]
        out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        N.check(N.lib().bsi_fourier_features(N.ptr(x), outer, C, inner, self.n_min, self.n_max, N.ptr(out),
                                             N.stream()))
        return out
