"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the in-kernel Gaussian noise generator of
`bsi_refine_step_philox` / `bsi_philox_normal` (bsi_amd/csrc/bsi_ops.hip), the opt-in replacement of the reference's
`torch.randn` draws in `BSI.sample` (/root/reference/bsi/bsi.py:325 mu_0 noise, :332-334 measurement noise).

The algorithm is not in the reference (which calls ATen's generator); it is the published counter-based generator
Philox4x32-10 of Salmon, Moraes, Dror, Shaw, "Parallel Random Numbers: As Easy as 1, 2, 3" (SC'11, section 3.3 and
table 2), as distributed in the Random123 library (philox.h: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key
increments 0x9E3779B9 / 0xBB67AE85).  Pinning: `philox4x32_10` below is checked against Random123's published
known-answer vectors (Random123 `examples/kat_vectors`, lines `philox4x32 10 ...`; copied as DATA into
tests/golden/philox_kat.json) by tests/test_philox_oracle.py on the CPU; the HIP kernel is checked against the same
vectors and against this restatement on the GPU (tests/test_hip_philox.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
STREAM_TAG = 0x42534931  # "BSI1": fourth counter word of the library's streams (bsi_ops.hip philox_normal4)
MU0_STREAM = 0xFFFFFFFF  # stream id of the mu_0 noise (bsi_amd/bsi.py _run_chain); measurement step i uses stream i


def philox4x32_10(counter, key):
    """counter [..., 4], key [..., 2] (integers < 2**32) -> [..., 4] uint32: ten rounds of
    (c0, c1, c2, c3) <- (hi(M1*c2) ^ c1 ^ k0, lo(M1*c2), hi(M0*c0) ^ c3 ^ k1, lo(M0*c0)), the key bumped by the
    Weyl constants between rounds (SC'11 section 3.3; Random123 philox.h `_philox4x32round` / `_philox4x32bumpkey`)."""
    c = np.asarray(counter, dtype=np.uint64) & MASK
    k = np.asarray(key, dtype=np.uint64) & MASK
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for r in range(10):
        p0 = M0 * c0  # < 2**64: both factors < 2**32
        p1 = M1 * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & MASK, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & MASK
        if r < 9:
            k0 = (k0 + np.uint64(W0)) & MASK
            k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def stream_uint32(seed: int, stream_id: int, n: int):
    """The library's counter / key layout: element group g (four consecutive elements) of noise stream `stream_id` is block
    Philox4x32-10(counter = (g & 0xffffffff, g >> 32, stream_id, STREAM_TAG), key = (seed & 0xffffffff, seed >> 32))."""
    assert n % 4 == 0
    g = np.arange(n // 4, dtype=np.uint64)
    ctr = np.stack([g & MASK, g >> np.uint64(32), np.full_like(g, stream_id & 0xFFFFFFFF), np.full_like(g, STREAM_TAG)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    return philox4x32_10(ctr, np.broadcast_to(key, (n // 4, 2))).reshape(-1)


def normals_from_uint32(x):
    """Box-Muller on 24-bit uniforms, in float64: words (x0, x1) -> u1 = ((x0 >> 8) + 1) / 2**24 in (0, 1],
    u2 = (x1 >> 8) / 2**24 in [0, 1) -> r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2); likewise (x2, x3)."""
    x = np.asarray(x, dtype=np.uint32).reshape(-1, 2)
    u1 = ((x[:, 0] >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    u2 = (x[:, 1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u1))
    return np.stack([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)], axis=-1).reshape(-1)


def normal_stream(seed: int, stream_id: int, n: int):
    """float64 values of what bsi_philox_normal(seed, stream_id, n) writes (the kernel evaluates the same formula in fp32 with
    the hardware's log2 / sin / cos units, which are not correctly rounded: compare with an absolute tolerance)."""
    return normals_from_uint32(stream_uint32(seed, stream_id, n))
