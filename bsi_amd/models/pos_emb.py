"""Mirror of bsi/models/pos_emb.py:7-84 on the native kernel `bsi_nyquist_embed`."""
import numpy as np
import torch
from torch import Tensor, nn

from .. import _native as N


def nyquist_tables(size: int, expected_rate: int):
    """(scale, bias) exactly as pos_emb.py:44-77: frequencies geomspace(1/8, nyquist/(2 phi), size/2) in
    float64, every frequency twice (second shifted by pi/2), stored as fp32."""
    assert size % 2 == 0
    k = size // 2
    nyquist_frequency = expected_rate / 2
    golden_ratio = (1 + np.sqrt(5)) / 2
    frequencies = np.geomspace(1 / 8, nyquist_frequency / (2 * golden_ratio), num=k)
    scale = np.repeat(2 * np.pi * frequencies, 2)
    bias = np.tile(np.array([0, np.pi / 2]), k)
    return torch.tensor(scale, dtype=torch.float32), torch.tensor(bias, dtype=torch.float32)


class NyquistPositionalEmbedding(nn.Module):
    """Sine-cosine embedding of t in [0, 1] with frequencies from 1/8 to Nyquist/(2 phi)."""

    @classmethod
    def from_config(cls, size, expected_rate, **kwargs):
        return cls(size, expected_rate)

    def __init__(self, size: int, expected_rate: int):
        super().__init__()
        self.size = size
        scale, bias = nyquist_tables(size, expected_rate)
        self.register_buffer("scale", scale, persistent=False)
        self.register_buffer("bias", bias, persistent=False)

    def forward(self, t: Tensor) -> Tensor:
        if t.dtype != torch.float32:
            raise RuntimeError("bsi_amd.NyquistPositionalEmbedding: the native kernel computes in fp32")
        tf = t.contiguous().reshape(-1)
        out = torch.empty((tf.numel(), self.size), dtype=torch.float32, device=t.device)
        N.check(N.lib().bsi_nyquist_embed(N.ptr(tf), tf.numel(), N.ptr(self.scale), N.ptr(self.bias), self.size,
                                          N.ptr(out), None, N.stream()))
        return out.reshape(*t.shape, self.size)
