"""SURVEY section 8(b) "Threading": the module must tolerate being traced.  The reference's task compiles the four entry points by
default (/root/reference/config/task/bsi.yaml:14 `compile: yes`, /root/reference/bsi/tasks/bsi.py:130-134:
`torch.compile(self.bsi.train_loss, mode=compile_mode)` and the same for elbo, sample, sample_history).  Here they are opaque to
dynamo (`torch.compiler.disable`, bsi_amd/bsi.py): the compiled wrappers must return the eager bits for the same generator state,
carry gradients, and compile nothing."""
import pytest
import torch

from tests.util import golden


def _tiny(dev):
    from bsi_amd import BSI, Discretization
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    from tests.util import weights
    model = DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8))
    model.load_state_dict(weights("dit_ff"))
    model = model.to(dev)
    return BSI(model, data_shape=(3, 16, 16), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=4, preconditioning="edm",
               discretization=Discretization.image_8bit()).to(dev)


def _nothing_traced(counters, n_wrappers):
    """No graph was built; the only "graph break" on record is dynamo skipping the disabled method; and one frame per compiled
    wrapper was examined ONCE -- repeated calls did not recompile."""
    assert counters["stats"]["unique_graphs"] == 0, dict(counters["stats"])
    assert all("torch.compiler.disable" in reason for reason in counters["graph_break"]), list(counters["graph_break"])
    assert counters["frames"]["total"] <= n_wrappers, dict(counters["frames"])


def test_entry_points_are_opaque_to_dynamo_cpu():
    """No GPU: a compiled entry point reaches the method body (which refuses CPU tensors -- there is no CPU path) without dynamo
    tracing anything."""
    import torch._dynamo
    from torch._dynamo.utils import counters
    from bsi_amd import BSI, Discretization
    torch._dynamo.reset()
    counters.clear()
    bsi = BSI(torch.nn.Conv2d(3, 3, 1), data_shape=(3, 8, 8), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=2, preconditioning="edm",
              discretization=Discretization.image_8bit())
    x = torch.zeros(2, 3, 8, 8)
    for fn, args in ((bsi.train_loss, (x,)), (bsi.elbo, (x, 1, 1)), (bsi.sample, (2,)), (bsi.sample_history, (2,))):
        with pytest.raises(RuntimeError, match="HIP device|CPU|cuda"):
            torch.compile(fn)(*args)
    _nothing_traced(counters, 4)


@pytest.mark.gpu
def test_compiled_entry_points_equal_eager_bits():
    import torch._dynamo
    from torch._dynamo.utils import counters
    dev = torch.device("cuda", 0)
    bsi = _tiny(dev)
    g4 = golden("g4_train_dit")
    x = g4["x"].to(dev)
    torch._dynamo.reset()
    counters.clear()
    c_train, c_elbo = torch.compile(bsi.train_loss), torch.compile(bsi.elbo)           # exactly as bsi/tasks/bsi.py:130-134
    c_sample, c_hist = torch.compile(bsi.sample), torch.compile(bsi.sample_history)    # (compile_mode: ~ = default mode)
    gen = lambda s: torch.Generator(dev).manual_seed(s)  # noqa: E731

    # train_loss + backward: eager vs compiled, twice each
    bsi.model.train()
    grads = []
    for fn in (bsi.train_loss, c_train, c_train):
        for p in bsi.model.parameters():
            p.grad = None
        loss = fn(x, gen(11))
        loss.mean().backward()
        grads.append((loss.detach().clone(), [p.grad.clone() for p in bsi.model.parameters()]))
    for loss, gs in grads[1:]:
        assert torch.equal(loss, grads[0][0])
        assert all(torch.equal(a, b) for a, b in zip(gs, grads[0][1]))
    bsi.model.eval()
    with torch.no_grad():
        e0 = bsi.elbo(x, 2, 2, gen(12), estimate_var=True)
        for _ in range(2):
            e1 = c_elbo(x, 2, 2, gen(12), estimate_var=True)
            assert torch.equal(e0[0], e1[0]) and torch.equal(e0[1], e1[1]) and torch.equal(e0[2]["bpd_var"], e1[2]["bpd_var"])
        s0 = bsi.sample(8, gen(13))
        h0 = bsi.sample_history(8, gen(14))
        for _ in range(2):
            assert torch.equal(s0, c_sample(8, gen(13)))
            h1 = c_hist(8, gen(14))
            assert all(torch.equal(a, b) for a, b in zip(h0, h1))
    # nothing was traced: no graphs, no graph breaks, hence no recompilation per call
    _nothing_traced(counters, 4)
