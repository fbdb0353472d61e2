"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares; the Python mirror has the reference's constructor signatures and state-dict contract; CPU
tensors are refused (no silent fallback)."""
import inspect
import os
import re

import pytest
import torch

from tests.util import golden, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bsi_amd import _native

    lib = _native.lib()
    hdr = open(os.path.join(ROOT, "include", "bsi_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|size_t|const char\*)\s+(bsi_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bsi_hip.h but not exported"
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    assert lib.bsi_version() >= 100


def test_cu_reserve_setter_contract():
    """bsi_set_cu_reserve (DPTrainer's CU budget for the RCCL kernels): 0 or a multiple of 8 up to 64; no GPU needed."""
    from bsi_amd import _native

    lib = _native.lib()
    assert lib.bsi_set_cu_reserve(0) == 0
    for bad in (-8, 4, 12, 72):
        assert lib.bsi_set_cu_reserve(bad) != 0
    assert lib.bsi_set_cu_reserve(16) == 0 and lib.bsi_set_cu_reserve(0) == 0


def test_launch_schedule_switches_are_per_thread():
    """bsi_set_cu_reserve / bsi_set_tile_queue / bsi_set_ln_stream_cus belong to the thread that launches: a training step that reserves
    CUs for its backward (on autograd's thread) must not change the grids of an evaluation on another thread.  No GPU needed:
    bsi_compute_cus() = CUs of the device (256 when none is visible) - the calling thread's reserve."""
    import threading

    from bsi_amd import _native

    lib = _native.lib()
    base = lib.bsi_compute_cus()
    seen = {}

    def other():
        seen["fresh"] = lib.bsi_compute_cus()          # the main thread's reserve of 16 is not visible here
        assert lib.bsi_set_cu_reserve(32) == 0
        seen["own"] = lib.bsi_compute_cus()

    try:
        assert lib.bsi_set_cu_reserve(16) == 0 and lib.bsi_compute_cus() == base - 16
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"fresh": base, "own": base - 32}, seen
        assert lib.bsi_compute_cus() == base - 16       # ... and the other thread's 32 is not visible here
    finally:
        assert lib.bsi_set_cu_reserve(0) == 0
    assert lib.bsi_compute_cus() == base


def test_bsi_surface_matches_reference():
    from bsi_amd import BSI, Discretization

    sig = inspect.signature(BSI.__init__)
    names = list(sig.parameters)
    assert names == ["self", "model", "data_shape", "lambda_0", "alpha_M", "alpha_R", "k", "preconditioning",
                     "low_discrepancy_sampling", "discretization"]
    assert all(sig.parameters[n].kind is inspect.Parameter.KEYWORD_ONLY for n in names[2:])
    m = torch.nn.Linear(2, 2)
    b = BSI(m, data_shape=(3, 8, 8), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=16, preconditioning="edm",
            discretization=Discretization.image_8bit())
    assert b.state_dict() == {} and list(b.children()) == []          # model is not a submodule
    assert b.model is m
    b.set_model(torch.nn.Identity())
    assert isinstance(b.model, torch.nn.Identity)
    assert b.lambda_0.ndim == 0 and b.default_schedule.shape == (17,)
    assert abs(b.p_lambda.ln_low - (-4.605170208339834)) < 1e-15
    assert abs(b.p_lambda.diff_ln_high_ln_low - 18.42068076630411) < 1e-13
    for meth in ("train_loss", "elbo", "finite_elbo", "sample", "sample_history", "reconstruction_loss",
                 "inf_measurement_loss", "finite_measurement_loss", "_predict_x", "_sample_q_mu_lambda",
                 "_edm_preconditioning", "_sample_lambda"):
        assert callable(getattr(b, meth))
    # no CPU path: CPU tensors raise instead of silently computing elsewhere
    with pytest.raises(RuntimeError):
        b.train_loss(torch.zeros(2, 3, 8, 8))
    with pytest.raises(RuntimeError):
        b.sample(2)


def test_discretization_known_answers():
    """The reference's own tests (tests/test_bsi.py:7-34) against the mirror class."""
    from bsi_amd import Discretization

    d = Discretization(0.0, 1.0, k=256)
    x = torch.tensor([-0.1, 0.0, 1.0, 1.0 - 1 / 256], dtype=torch.float64)
    assert d.bucketize(x).tolist() == [0, 0, 255, 254]
    d = Discretization(-1.0, 1.0, k=5)
    b = d.bin_boundaries(torch.device("cpu"), torch.float64)
    assert d.bucketize(b)[:-1].tolist() == list(range(5))
    assert d.bucketize(b - 1e-8)[1:].tolist() == list(range(5))
    d = Discretization(-1.0, 1.0, k=3)
    assert torch.allclose(d.bin_boundaries(torch.device("cpu"), torch.float32),
                          torch.tensor([-1.5, -0.5, 0.5, 1.5]))
    g = golden("kat_reference_tests")
    img = Discretization.image_8bit().to_8bit_image(torch.tensor([-1.2, -1.0, -0.5, 0.0, 0.999, 1.0, 1.5]))
    assert torch.equal(img, g["img8"])


def test_dit_state_dict_contract():
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures

    W = weights("dit_ff")
    m = DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8),
                     name="dit")
    sd = m.state_dict()
    assert set(sd) == set(W), set(sd) ^ set(W)
    assert all(sd[k].shape == W[k].shape for k in W)
    m.load_state_dict(W)  # strict
    g = golden("g7_components")
    assert torch.equal(m.dit.patch_pos_embedding, g["dit16_pos"])
    # zero-initialised adaLN output layer (dit.py:84-85)
    fresh = DenoisingDiT((3, 16, 16), 2, 128, 1, 2)
    assert float(fresh.dit.blocks[0].adaLN_modulation[2].weight.detach().abs().sum()) == 0.0
    assert fresh.dit.patch_encoder.weight.shape == (128, 12)
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 3, 16, 16), torch.zeros(1))  # CPU parameters: no CPU path


def test_nyquist_tables_match_reference():
    from bsi_amd.models.pos_emb import NyquistPositionalEmbedding

    g = golden("g7_components")
    for size, rate in [(1024, 1000), (32, 100), (512, 32), (64, 16)]:
        pe = NyquistPositionalEmbedding.from_config(size, rate, name="nyquist")
        assert torch.equal(pe.scale, g[f"pe_{size}_{rate}_scale"])
        assert torch.equal(pe.bias, g[f"pe_{size}_{rate}_bias"])
        assert pe.state_dict() == {}


def test_unet_state_dict_contract():
    from bsi_amd.models.pos_emb import NyquistPositionalEmbedding
    from bsi_amd.models.vdm_unet import DenoisingVDMUNet
    from bsi_amd.nn import FourierFeatures

    W = weights("unet_ff")
    m = DenoisingVDMUNet((3, 8, 8), NyquistPositionalEmbedding(32, 100), "silu", 64, 1, 4, n_attention_heads=1,
                         dropout=0.1, downsampling_attention=False, fourier_features=FourierFeatures(n_min=6, n_max=8),
                         name="unet")
    sd = m.state_dict()
    assert set(sd) == set(W), set(sd) ^ set(W)
    assert all(sd[k].shape == W[k].shape for k in W)
    m.load_state_dict(W)
    # without a Dropout module the second conv moves from layers.6 to layers.5
    m2 = DenoisingVDMUNet((3, 8, 8), NyquistPositionalEmbedding(32, 100), "silu", 64, 1, 4, dropout=None)
    assert "u_net.downsampling_blocks.0.0.layers.5.weight" in m2.state_dict()
    assert "u_net.upsampling_blocks.0.0.skip.weight" in m2.state_dict()
    assert m2.encode.weight.shape == (64, 3, 3, 3)
