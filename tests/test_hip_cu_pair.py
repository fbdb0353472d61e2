"""The two-resource evaluation (bsi_dit_forward_pair: two half-batch chains interleaved over a pair of CU-masked streams) must return
the BITS of the one-stream engine: every image is an independent chain in the reference (bsi/bsi.py:312-336, bsi/models/dit.py:96-103)
and the pair only changes which CUs run which launch and in which order the two halves' launches meet.  Checked on a small model
(odd batch, per-image conditioning, both placements of attention, tile queue on and off) and at the benchmark's size (DiT-L/2,
512 images = 256 + 256) for one evaluation and a 4-step sampling chain."""
import pytest
import torch

from tests.util import report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bsi(model, shape, k):
    from bsi_amd import BSI, Discretization
    return BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k, preconditioning="edm",
               discretization=Discretization.image_8bit()).to(DEV)


def _dit(shape, patch, dim, depth, heads):
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    torch.manual_seed(0)
    m = DenoisingDiT(shape, patch, dim, depth, heads, dropout=0.05, fourier_features=FourierFeatures(n_min=6, n_max=8))
    with torch.no_grad():
        for blk in m.dit.blocks:
            blk.adaLN_modulation[-1].weight.normal_(0, 0.02)
    m.cu_pair = None
    return m.to(DEV).eval()


@pytest.mark.parametrize("B", [64, 97, 200])
def test_pair_equals_one_stream_small_model(B):
    from bsi_amd import _native as N
    shape = (3, 32, 32)
    model = _dit(shape, 2, 128, 3, 2)
    bsi = _bsi(model, shape, 4)
    gen = torch.Generator(DEV).manual_seed(5)
    mu = torch.randn((B, *shape), device=DEV, generator=gen) * 2
    t = torch.rand(B, device=DEV, generator=gen)
    lib = N.lib()
    try:
        with torch.no_grad():
            ref = bsi._predict_x(mu, t)                       # per-image conditioning rows and coefficients (train_loss / elbo path)
            ref_s = bsi.sample(B, torch.Generator(DEV).manual_seed(9))   # shared row (sampling path)
            for h_cus in (8, 32):
                for flags in (0, 1):
                    for queue in (0, 1):
                        N.check(lib.bsi_set_tile_queue(queue))
                        model.cu_pair = (h_cus, flags)
                        got = bsi._predict_x(mu, t)
                        got_s = bsi.sample(B, torch.Generator(DEV).manual_seed(9))
                        model.cu_pair = None
                        assert torch.equal(got, ref), f"pair(h={h_cus}, flags={flags}, queue={queue}) evaluation differs"
                        assert torch.equal(got_s, ref_s), f"pair(h={h_cus}, flags={flags}, queue={queue}) sampling chain differs"
    finally:
        N.check(lib.bsi_set_tile_queue(0))
        model.cu_pair = None
    assert lib.bsi_compute_cus() == torch.cuda.get_device_properties(0).multi_processor_count, "the pair left a CU reserve behind"


def test_pair_equals_one_stream_full_size():
    """DiT-L/2 at the benchmark's batch: 256 + 256 over the pair vs 512 on one stream, one evaluation and a 4-step chain."""
    from bsi_amd import _native as N
    shape, B = (3, 32, 32), 512
    model = _dit(shape, 2, 1024, 24, 16)
    bsi = _bsi(model, shape, 4)
    gen = torch.Generator(DEV).manual_seed(21)
    mu = torch.randn((B, *shape), device=DEV, generator=gen) * 2
    t = torch.rand(B, device=DEV, generator=gen)
    lib = N.lib()
    try:
        with torch.no_grad():
            ref = bsi._predict_x(mu, t)
            ref_s = bsi.sample(B, torch.Generator(DEV).manual_seed(3))
            assert bool(torch.isfinite(ref).all()) and bool(torch.isfinite(ref_s).all())
            for h_cus, flags, queue in ((24, 0, 1), (32, 1, 0), (16, 0, 0)):
                N.check(lib.bsi_set_tile_queue(queue))
                model.cu_pair = (h_cus, flags)
                got = bsi._predict_x(mu, t)
                got_s = bsi.sample(B, torch.Generator(DEV).manual_seed(3))
                model.cu_pair = None
                assert torch.equal(got, ref), f"pair(h={h_cus}, flags={flags}, queue={queue}) evaluation differs"
                assert torch.equal(got_s, ref_s), f"pair(h={h_cus}, flags={flags}, queue={queue}) sampling chain differs"
    finally:
        N.check(lib.bsi_set_tile_queue(0))
        model.cu_pair = None
    report("cu_pair_fullsize", model="dit_l2", batch=B, halves="256+256", bit_exact=True)


def test_pair_rejects_bad_partitions():
    import ctypes as C
    from bsi_amd import _native as N
    lib = N.lib()
    h = C.c_void_p()
    for bad in (0, 4, 12, 72, 256):
        assert lib.bsi_cu_pair_create(bad, C.byref(h)) != 0 and not h.value
