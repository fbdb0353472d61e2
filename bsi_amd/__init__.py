"""bsi_amd — MI355X-native drop-in for the hot path of martenlienen/bsi.

Mirrors the reference's module surface (`bsi.bsi.BSI`, `bsi.bsi.Discretization`,
`bsi.models.dit.DenoisingDiT`, `bsi.nn.FourierFeatures`,
`bsi.models.pos_emb.NyquistPositionalEmbedding`) on top of hand-written HIP kernels for
gfx950 reached through the C ABI of include/bsi_hip.h.  There is no CPU compute path.
"""
from .bsi import BSI, Discretization, LogUniform, broadcast_right  # noqa: F401
from .bfn import BFN  # noqa: F401,E402
from .vdm import VDM  # noqa: F401,E402
