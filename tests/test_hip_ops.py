"""GPU parity tests of the individual HIP kernels (called through the C ABI) against the CPU oracle and
the golden vectors.  Tolerances are written next to each check: fp32 elementwise kernels match the
oracle to a few ulp; bf16-MFMA kernels are compared with an oracle fed the same bf16-rounded operands."""
import ctypes as C
import math

import pytest
import torch

from oracle import bsi_oracle as bo
from oracle import dit_oracle as do
from tests.util import golden, max_rel, rel_linf, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def N():
    from bsi_amd import _native
    _native.lib()
    return _native


def params(N):
    o = bo.BSIOracle(None, data_shape=(3, 8, 8))
    return N.BSIParams(0.01, 1e6, 2e6, o.p_lambda.ln_low, o.p_lambda.delta), o


_KEEP = []  # device copies stay alive until the test ends (kernels are enqueued asynchronously)


@pytest.fixture(autouse=True)
def _release_device_copies():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def dev(t):
    d = t.to(DEV).contiguous()
    _KEEP.append(d)
    return d


def empty(*shape, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=DEV)


# ----------------------------------------------------------------------------------------------
# BSI wrapper kernels
# ----------------------------------------------------------------------------------------------
def test_edm_coeffs_and_schedule(N):
    g = golden("g1_tables")
    p, o = params(N)
    t = dev(g["t"])
    n = t.numel()
    lam, cs, co, ci = empty(n), empty(n), empty(n), empty(n)
    N.check(N.lib().bsi_edm_coeffs(C.byref(p), N.ptr(t), n, N.ptr(lam), N.ptr(cs), N.ptr(co), N.ptr(ci), N.stream()))
    # fp32, same op order as the reference: differences come only from expf/sqrtf ulp -> 1e-6 relative
    assert max_rel(lam, g["lam"]) < 2e-6
    assert max_rel(co, g["c_out"]) < 4e-6 and max_rel(ci, g["c_in"]) < 4e-6
    assert float((cs.cpu() - g["c_skip"]).abs().max()) < 4e-6 * float(g["c_skip"].abs().max())
    tt, rp = empty(n), empty(n)
    N.check(N.lib().bsi_lambda_to_t(C.byref(p), N.ptr(dev(g["lam"])), n, N.ptr(tt), N.ptr(rp), N.stream()))
    assert float((tt.cpu() - g["cdf_lam"]).abs().max()) < 2e-7 and max_rel(rp, g["rpdf"]) < 1e-7
    ts = dev(g["default_schedule"])
    k1 = ts.numel()
    l2, al = empty(k1), empty(k1 - 1)
    N.check(N.lib().bsi_schedule(C.byref(p), N.ptr(ts), k1, N.ptr(l2), N.ptr(al), N.stream()))
    lam_ref = o.p_lambda.icdf(g["default_schedule"])
    assert max_rel(l2, lam_ref) < 2e-6
    assert rel_linf(al, lam_ref.diff()) < 2e-6
    assert torch.equal(al.cpu(), l2.cpu().diff())  # alpha is exactly the fp32 diff of the kernel's own lambdas


def test_lambda_grid_and_q_sample(N):
    g = golden("g2g3_lambda_q")
    p, o = params(N)
    for n, B in [(1, 8), (3, 5)]:
        lam = empty(n * B)
        N.check(N.lib().bsi_lambda_grid(C.byref(p), N.ptr(dev(g[f"perm_{n}_{B}"])), N.ptr(dev(g[f"offset_{n}_{B}"])),
                                        n * B, N.ptr(lam), N.stream()))
        assert max_rel(lam.reshape(n, B), g[f"lam_{n}_{B}"]) < 4e-6
    x, lam, eps = dev(g["x"]), dev(g["q_lam"]), dev(g["q_eps"])
    mu = torch.empty_like(eps)
    N.check(N.lib().bsi_q_sample(C.byref(p), N.ptr(x), N.ptr(lam), N.ptr(eps), 15, 5, 192, N.ptr(mu), N.stream()))
    assert torch.equal(mu.cpu(), g["q_mu"])  # same fp32 op sequence, exact div/sqrt: bit-exact


def test_refine_step_and_combine(N):
    p, o = params(N)
    gen = torch.Generator().manual_seed(3)
    n, D, k = 6, 3 * 32 * 32, 16
    t = o.default_schedule if o.k == k else torch.linspace(0, 1, k + 1)
    lam = o.p_lambda.icdf(t)
    alpha = lam.diff()
    cs, co, ci = o.edm_coeffs(t)
    mu = torch.randn((n, D), generator=gen) * 3
    f = torch.randn((n, D), generator=gen)
    eps = torch.randn((n, D), generator=gen)
    dmu, df, deps, dlam, dal, dcs, dco = map(dev, (mu, f, eps, lam, alpha, cs, co))
    for i in (0, 7, 15):
        xh = torch.addcmul(cs[i] * mu, co[i], f)
        y = xh + torch.rsqrt(alpha[i]) * eps
        mn = (alpha[i] * y + lam[i] * mu) / lam[i + 1]
        oxh, oy, omn = empty(n, D), empty(n, D), empty(n, D)
        N.check(N.lib().bsi_refine_step(N.ptr(dmu), N.ptr(df), N.ptr(deps), N.ptr(dlam), N.ptr(dal), N.ptr(dcs),
                                        N.ptr(dco), i, 0, n, D, N.ptr(oxh), N.ptr(oy), N.ptr(omn), N.stream()))
        assert torch.equal(oxh.cpu(), xh) and torch.equal(oy.cpu(), y) and torch.equal(omn.cpu(), mn)
        # f already an x_hat, no history outputs
        omn2 = empty(n, D)
        N.check(N.lib().bsi_refine_step(N.ptr(dmu), N.ptr(dev(xh)), N.ptr(deps), N.ptr(dlam), N.ptr(dal), None, None,
                                        i, 1, n, D, None, None, N.ptr(omn2), N.stream()))
        assert torch.equal(omn2.cpu(), mn)
    # per-row coefficients
    csr, cor, cir = (torch.rand(n, generator=gen) for _ in range(3))
    out = empty(n, D)
    N.check(N.lib().bsi_predict_combine(N.ptr(dmu), N.ptr(df), N.ptr(dev(csr)), N.ptr(dev(cor)), 1, n, D, N.ptr(out),
                                        N.stream()))
    assert torch.equal(out.cpu(), torch.addcmul(csr[:, None] * mu, cor[:, None], f))
    N.check(N.lib().bsi_scale_rows(N.ptr(dmu), N.ptr(dev(cir)), 1, n, D, N.ptr(out), N.stream()))
    assert torch.equal(out.cpu(), cir[:, None] * mu)
    N.check(N.lib().bsi_sample_init(N.ptr(deps), N.ptr(dlam), n, D, N.ptr(out), N.stream()))
    assert torch.equal(out.cpu(), torch.rsqrt(lam[0]) * eps)
    gf, gm = empty(n, D), empty(n, D)
    N.check(N.lib().bsi_predict_combine_bwd(N.ptr(df), N.ptr(dev(csr)), N.ptr(dev(cor)), 1, n, D, N.ptr(gf), N.ptr(gm),
                                            N.stream()))
    assert torch.equal(gf.cpu(), cor[:, None] * f) and torch.equal(gm.cpu(), csr[:, None] * f)


def test_philox_device_noise(N):
    """In-kernel Gaussian noise (opt-in): moments of the generator, independence of the streams, and the fused step
    bsi_refine_step_philox == bsi_refine_step fed with the same stream written to memory by bsi_philox_normal (bit for bit)."""
    P, o = params(N)
    n = 1 << 20
    seed = dev(torch.tensor([0x1234_5678_9ABC_DEF], dtype=torch.int64))
    a, b, a2 = empty(n), empty(n), empty(n)
    N.check(N.lib().bsi_philox_normal(N.ptr(seed), 3, n, N.ptr(a), N.stream()))
    N.check(N.lib().bsi_philox_normal(N.ptr(seed), 4, n, N.ptr(b), N.stream()))
    N.check(N.lib().bsi_philox_normal(N.ptr(seed), 3, n, N.ptr(a2), N.stream()))
    assert torch.equal(a, a2) and not torch.equal(a, b)
    for x in (a.double().cpu(), b.double().cpu()):
        assert abs(float(x.mean())) < 5e-3 and abs(float(x.var()) - 1) < 1e-2
        assert abs(float((x ** 3).mean())) < 2e-2 and abs(float((x ** 4).mean()) - 3) < 6e-2
        assert float(x.abs().max()) < 6.0 and float((x.abs() > 3).double().mean()) == pytest.approx(0.0027, abs=5e-4)
    assert abs(float((a.double() * b.double()).mean())) < 5e-3                       # streams are uncorrelated
    assert abs(float((a[0::4].double() * a[1::4].double()).mean())) < 1e-2           # and so are a group's four normals
    other = dev(torch.tensor([77], dtype=torch.int64))
    N.check(N.lib().bsi_philox_normal(N.ptr(other), 3, n, N.ptr(a2), N.stream()))
    assert not torch.equal(a, a2)
    # fused step
    gen = torch.Generator().manual_seed(5)
    rows, D, k, i = 6, 192, 8, 3
    mu, f = dev(torch.randn((rows, D), generator=gen)), dev(torch.randn((rows, D), generator=gen))
    t = torch.linspace(0, 1, k + 1)
    lam = o.p_lambda.icdf(t)
    alpha = lam.diff()
    cs, co, ci = o.edm_coeffs(t)
    eps = empty(rows, D)
    N.check(N.lib().bsi_philox_normal(N.ptr(seed), i, rows * D, N.ptr(eps), N.stream()))
    outs = []
    for philox in (False, True):
        xh, y, mn = empty(rows, D), empty(rows, D), empty(rows, D)
        fn = N.lib().bsi_refine_step_philox if philox else N.lib().bsi_refine_step
        N.check(fn(N.ptr(mu), N.ptr(f), N.ptr(seed) if philox else N.ptr(eps), N.ptr(dev(lam)), N.ptr(dev(alpha)), N.ptr(dev(cs)),
                   N.ptr(dev(co)), i, 0, rows, D, N.ptr(xh), N.ptr(y), N.ptr(mn), N.stream()))
        outs.append((xh, y, mn))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


def test_sqerr_and_backward(N):
    gen = torch.Generator().manual_seed(4)
    B, n, D = 5, 3, 3072
    x = torch.rand((B, D), generator=gen) * 2 - 1
    xh = x.repeat(n, 1) + 0.01 * torch.randn((n * B, D), generator=gen)
    w = torch.rand(n * B, generator=gen) * 1e6
    out = empty(n * B)
    N.check(N.lib().bsi_sqerr_rows(N.ptr(dev(x)), N.ptr(dev(xh)), N.ptr(dev(w)), 0.5, 0, n * B, B, D, N.ptr(out), N.stream()))
    ref = 0.5 * w.double() * (x.double().repeat(n, 1) - xh.double()).square().sum(1)
    assert max_rel(out, ref) < 2e-6  # fp32 tree reduction of 3072 terms vs fp64
    N.check(N.lib().bsi_sqerr_rows(N.ptr(dev(x)), N.ptr(dev(xh)), N.ptr(dev(w)), 1.0, 1, n * B, B, D, N.ptr(out), N.stream()))
    assert max_rel(out, w.double() * (x.double().repeat(n, 1) - xh.double()).square().mean(1)) < 2e-6
    g = torch.rand(n * B, generator=gen)
    gx = empty(n * B, D)
    N.check(N.lib().bsi_sqerr_rows_bwd(N.ptr(dev(x)), N.ptr(dev(xh)), N.ptr(dev(w)), N.ptr(dev(g)), 1.0, 1, n * B, B, D,
                                       N.ptr(gx), N.stream()))
    xr = xh.clone().requires_grad_(True)
    (w * (x.repeat(n, 1) - xr).square().mean(1) * g).sum().backward()
    assert rel_linf(gx, xr.grad) < 1e-6


def test_recon_nll_and_uint8(N):
    g = golden("g6_elbo")
    d = bo.Disc.image_8bit()
    x = g["x"]
    B, D = x.shape[0], 192
    W = sub(g, "W.")

    def f(mu, t):
        tp = t.reshape(-1, 1, 1, 1).expand(-1, 1, 8, 8)
        return torch.nn.functional.conv2d(torch.cat((mu, tp), 1), W["layer.weight"], W["layer.bias"], padding=1)

    o = bo.BSIOracle(f, data_shape=(3, 8, 8), k=16, discretization=d)
    eps = g["eps_r"]
    n = eps.shape[0]
    lam_M = x.new_full((n, B), 1e6)
    mu = o.q_mu_lambda(x, lam_M, eps).flatten(end_dim=1)
    with torch.no_grad():
        xh = o.predict_x(mu, x.new_ones(n * B))
    out = empty(n * B)
    bounds = d.bin_boundaries(torch.float32)
    N.check(N.lib().bsi_recon_nll(N.ptr(dev(x)), N.ptr(dev(xh)), 2e6, N.ptr(dev(bounds)), d.lo - d.dx / 2, d.dx, d.k,
                                  n * B, B, D, N.ptr(out), N.stream()))
    # erf/log ulp differences get amplified by the cdf difference; 1e-4 relative on the 192-term sum
    assert max_rel(out.reshape(n, B), g["l_recon"]) < 1e-4
    N.check(N.lib().bsi_recon_nll(N.ptr(dev(x)), N.ptr(dev(xh)), 2e6, None, 0.0, 1.0, 0, n * B, B, D, N.ptr(out),
                                  N.stream()))
    oc = bo.BSIOracle(f, data_shape=(3, 8, 8), k=16, discretization=None)
    sigma = torch.rsqrt(oc.alpha_R)
    ref = -(-((x - xh.reshape(n, B, 3, 8, 8)) ** 2) / (2 * sigma ** 2) - torch.log(sigma)
            - math.log(math.sqrt(2 * math.pi))).reshape(n, B, -1).sum(2)
    assert max_rel(out.reshape(n, B), ref) < 1e-5
    v = torch.linspace(-1.3, 1.3, 4001)
    u8 = torch.empty(4001, dtype=torch.uint8, device=DEV)
    N.check(N.lib().bsi_to_uint8(N.ptr(dev(v)), -1.0, 1.0, 4001, N.ptr(u8), N.stream()))
    assert torch.equal(u8.cpu(), d.to_8bit_image(v))


# ----------------------------------------------------------------------------------------------
# DiT building blocks
# ----------------------------------------------------------------------------------------------
def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def test_embeddings_and_fourier_features(N):
    g = golden("g7_components")
    for size, rate in [(1024, 1000), (32, 100), (512, 32), (64, 16)]:
        t = g[f"pe_{size}_{rate}_t"]
        out = empty(t.numel(), size)
        ob = empty(t.numel(), size, dtype=torch.bfloat16)
        N.check(N.lib().bsi_nyquist_embed(N.ptr(dev(t)), t.numel(), N.ptr(dev(g[f"pe_{size}_{rate}_scale"])),
                                          N.ptr(dev(g[f"pe_{size}_{rate}_bias"])), size, N.ptr(out), N.ptr(ob), N.stream()))
        # same fp32 argument (fma), accurate sinf on both sides: <= 2 ulp of values in [-1,1]
        assert float((out.cpu() - g[f"pe_{size}_{rate}_out"]).abs().max()) < 3e-7
        assert torch.equal(ob.cpu(), out.cpu().to(torch.bfloat16))
    x = g["ff_x"]
    out = empty(*g["ff_out"].shape)
    N.check(N.lib().bsi_fourier_features(N.ptr(dev(x)), 2, 3, 16, 6, 8, N.ptr(out), N.stream()))
    assert float((out.cpu() - g["ff_out"]).abs().max()) < 3e-7
    # the reference's own known-answer test (test_fourier_features.py:9-28) through the mirror module
    from bsi_amd.nn import FourierFeatures
    import numpy as np
    m = FourierFeatures(n_min=5, n_max=6).to(DEV)
    xv = torch.tensor([1.333, -np.e / 7], dtype=torch.float32)[None, :, None].repeat(2, 1, 3)
    y = m(xv.to(DEV), dim=1).cpu()
    assert m.n_features() == 4 and y.shape == (2, 8, 3)
    exp64 = do.fourier_features(xv.double(), 5, 6, dim=1)  # fp32 tables, fp64 evaluation
    # fp32 evaluation of sin at |arg| ~ 500 with fp32 argument rounding: 3e-5 absolute
    assert float((y.double() - exp64).abs().max()) < 6e-5


@pytest.mark.parametrize("M,Nn,K", [(256, 256, 64), (512, 1024, 1024), (300, 384, 128), (64, 768, 512),
                                     (1, 1024, 1024), (129, 6144, 1024), (1024, 1024, 4096), (2048, 4096, 1024)])
def test_gemm_epilogues(N, M, Nn, K):
    gen = torch.Generator().manual_seed(M * 7 + Nn + K)
    A = bf16r(torch.randn((M, K), generator=gen))
    W = bf16r(torch.randn((Nn, K), generator=gen) / math.sqrt(K))
    bias = torch.randn(Nn, generator=gen)
    ref = A.double() @ W.double().t() + bias.double()
    dA, dW, db = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(bias)

    def run(epi, out, **kw):
        a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K,
                       lda=K, ldw=K, ldo=Nn, epilogue=epi, **kw)
        N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
        return out

    # fp32 accumulation of exact bf16 products: error ~ K * 2^-24 relative to sum|a*b| -> 1e-5 of the output scale
    o = run(N.EPI_BIAS_F32, empty(M, Nn))
    assert rel_linf(o, ref) < 2e-5, rel_linf(o, ref)
    # bf16 outputs: one rounding (2^-9 relative) on top
    o = run(N.EPI_BIAS_BF16, empty(M, Nn, dtype=torch.bfloat16))
    assert float(((o.cpu().double() - ref).abs() / (ref.abs() + 1e-2)).max()) < 5e-3
    o = run(N.EPI_BIAS_GELU_BF16, empty(M, Nn, dtype=torch.bfloat16))
    assert float((o.cpu().double() - do.gelu_tanh(ref)).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()) / 4)
    assert rel_linf(o.cpu().float(), do.gelu_tanh(ref)) < 5e-3
    o = run(N.EPI_BIAS_SILU_BF16, empty(M, Nn, dtype=torch.bfloat16))
    assert rel_linf(o.cpu().float(), do.silu(ref)) < 5e-3
    tokens = 64 if M % 64 == 0 else 1
    rows = max(M // tokens, 1)
    for gate_rows in {1, rows}:
        gate = torch.randn((gate_rows, 3 * Nn), generator=gen)
        x0 = torch.randn((M, Nn), generator=gen)
        xg = dev(x0)
        dgate = dev(gate)
        run(N.EPI_GATE_RESID, xg, gate=dgate.data_ptr() + 4 * Nn, gate_rows=gate_rows, gate_stride=3 * Nn,
            tokens=tokens)
        gsel = gate[(torch.arange(M) // tokens) % gate_rows, Nn:2 * Nn]
        assert rel_linf(xg, x0.double() + gsel.double() * ref) < 2e-5
    pos = torch.randn((tokens, Nn), generator=gen)
    dpos = dev(pos)
    o = run(N.EPI_BIAS_POS_F32, empty(M, Nn), pos=dpos.data_ptr(), tokens=tokens)
    assert rel_linf(o, ref + pos[torch.arange(M) % tokens].double()) < 2e-5


@pytest.mark.parametrize("M,Nn,K", [(512, 256, 256), (1000, 320, 192), (4096 + 40, 1024, 256), (65536, 768, 128), (200, 512, 64)])
def test_gemm_training_epilogues(N, M, Nn, K):
    """The two training epilogues of the MLP (dit.py:71-76 forward with the pre-activation saved for autograd; backward of the
    GELU fused into the fc2 input gradient): BIAS_GELU_DUAL writes bf16(acc + bias) and bf16(gelu(acc + bias)), MUL_GELUGRAD
    writes bf16((acc + bias) * gelu'(aux)).  Full tiles, ragged M / N tails (the counted-wait and the drain form of the
    auxiliary-row pipeline) and many tiles per workgroup."""
    gen = torch.Generator().manual_seed(M + 3 * Nn + K)
    A = bf16r(torch.randn((M, K), generator=gen))
    W = bf16r(torch.randn((Nn, K), generator=gen) / math.sqrt(K))
    bias = torch.randn(Nn, generator=gen)
    aux = bf16r(torch.randn((M, Nn), generator=gen) * 1.5)
    ref = A.double() @ W.double().t() + bias.double()
    dA, dW, db, dx = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(bias), dev(aux.to(torch.bfloat16))
    out = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
    out2 = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
    a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), out2=out2.data_ptr(), M=M, N=Nn, K=K,
                   lda=K, ldw=K, ldo=Nn, epilogue=N.EPI_BIAS_GELU_DUAL)
    N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
    assert float(((out2.cpu().double() - ref).abs() / (ref.abs() + 1e-2)).max()) < 5e-3
    # the activation is applied to the fp32 value, not to the rounded pre-activation
    assert rel_linf(out.cpu().float(), do.gelu_tanh(ref)) < 5e-3
    x = aux.double().requires_grad_(True)
    do.gelu_tanh(x).sum().backward()
    want = ref * x.grad
    out.fill_(float("nan"))
    a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), aux=dx.data_ptr(), M=M, N=Nn, K=K,
                   lda=K, ldw=K, ldo=Nn, epilogue=N.EPI_MUL_GELUGRAD_BF16)
    N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
    assert rel_linf(out.cpu().float(), want) < 5e-3, rel_linf(out.cpu().float(), want)
    # and twice the same bits (no data race on the wait counts)
    first = out.clone()
    N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
    assert torch.equal(first.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize("M,Nn,K", [(512, 256, 256), (1000, 320, 192), (4096 + 40, 1024, 256)])
def test_gemm_gelugrad_column_sums_and_row_table_reduction(N, M, Nn, K):
    """bsi_gemm_args::colsum_rows: the MUL_GELUGRAD epilogue leaves, per 128-row slab, the column sums of the bf16 values it stores (the
    fc1 bias gradient, autograd's grad_output.sum(0), taken where grad_output is produced), and bsi_colsum_rows_f32 adds the slabs
    in fixed order.  The slab table must equal the sums of the OUTPUT's rows (fp32 adds of at most 128 values), the reduced vector
    the column sum of the whole output; ragged M / N tails included; the main output is bit-identical with and without the rider."""
    gen = torch.Generator().manual_seed(M + 5 * Nn + K)
    A = bf16r(torch.randn((M, K), generator=gen))
    W = bf16r(torch.randn((Nn, K), generator=gen) / math.sqrt(K))
    aux = bf16r(torch.randn((M, Nn), generator=gen) * 1.5)
    dA, dW, dx = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(aux.to(torch.bfloat16))
    out = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
    plain = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
    slabs = (M + 127) // 128
    rows = torch.full((slabs, Nn), float("nan"), dtype=torch.float32, device=DEV)
    a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=None, out=plain.data_ptr(), aux=dx.data_ptr(), M=M, N=Nn, K=K,
                   lda=K, ldw=K, ldo=Nn, epilogue=N.EPI_MUL_GELUGRAD_BF16)
    N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
    a.out, a.colsum_rows = out.data_ptr(), rows.data_ptr()
    N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
    assert torch.equal(out.view(torch.int16), plain.view(torch.int16))
    o = out.cpu().double()
    want_rows = torch.stack([o[128 * i:128 * (i + 1)].sum(0) for i in range(slabs)])
    scale = o.abs().sum(0).max()
    assert float((rows.cpu().double() - want_rows).abs().max()) < 2e-6 * float(scale)
    res = torch.full((Nn,), float("nan"), dtype=torch.float32, device=DEV)
    scratch = torch.empty(max(1, N.lib().bsi_colsum_rows_scratch_bytes(slabs, Nn)), dtype=torch.uint8, device=DEV)
    job = N.ColsumJob(src=rows.data_ptr(), rows=slabs, cols=Nn, ld=Nn, out=res.data_ptr())
    N.check(N.lib().bsi_colsum_rows_f32(C.byref(job), 1, N.ptr(scratch), N.stream()))
    assert float((res.cpu().double() - o.sum(0)).abs().max()) < 2e-6 * float(scale)


def test_colsum_rows_three_jobs_two_stages(N):
    """bsi_colsum_rows_f32: three tables in one pair of launches -- one above 64 rows (two stages, ragged last chunk), one of a
    single chunk (written directly), one with a leading dimension wider than its columns -- against fp64, and twice the same bits."""
    gen = torch.Generator().manual_seed(11)
    t0 = dev(torch.randn((2048 + 37, 1024), generator=gen))
    t1 = dev(torch.randn((40, 260), generator=gen))
    t2 = dev(torch.randn((130, 512), generator=gen))  # columns 0..299 of a 512-wide table
    outs = [torch.full((n,), float("nan"), dtype=torch.float32, device=DEV) for n in (1024, 260, 300)]
    jobs = (N.ColsumJob * 3)(N.ColsumJob(src=t0.data_ptr(), rows=t0.shape[0], cols=1024, ld=1024, out=outs[0].data_ptr()),
                             N.ColsumJob(src=t1.data_ptr(), rows=40, cols=260, ld=260, out=outs[1].data_ptr()),
                             N.ColsumJob(src=t2.data_ptr(), rows=130, cols=300, ld=512, out=outs[2].data_ptr()))
    need = sum(N.lib().bsi_colsum_rows_scratch_bytes(r, c) for r, c in ((t0.shape[0], 1024), (40, 260), (130, 300)))
    scratch = torch.empty(need, dtype=torch.uint8, device=DEV)
    N.check(N.lib().bsi_colsum_rows_f32(jobs, 3, N.ptr(scratch), N.stream()))
    first = [o.clone() for o in outs]
    for o, t, c in zip(outs, (t0, t1, t2), (1024, 260, 300)):
        assert rel_linf(o, t.cpu().double()[:, :c].sum(0)) < 2e-6
    N.check(N.lib().bsi_colsum_rows_f32(jobs, 3, N.ptr(scratch), N.stream()))
    for o, f in zip(outs, first):
        assert torch.equal(o, f)
    assert N.lib().bsi_colsum_rows_f32(jobs, 5, N.ptr(scratch), N.stream()) != 0


@pytest.mark.parametrize("epi_name", ["bias", "gelu"])
def test_gemm_persistent_tile_handover_full_size(N, epi_name):
    """768 tiles on 256 CUs: every persistent workgroup walks 3 tiles, so the DMA ring, the epilogue's store allowance and the
    half-stage issued in front of the epilogue are exercised across tile boundaries.  Production schedule (variant 12) against
    the K = 32 ring (variant 6) on the whole output, and against fp64 on sampled rows."""
    M, Nn, K = 16384, 3072, 1024
    gen = torch.Generator().manual_seed(11)
    A = bf16r(torch.randn((M, K), generator=gen))
    W = bf16r(torch.randn((Nn, K), generator=gen) / math.sqrt(K))
    bias = torch.randn(Nn, generator=gen)
    dA, dW, db = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(bias)
    epi = N.EPI_BIAS_BF16 if epi_name == "bias" else N.EPI_BIAS_GELU_BF16
    outs = {}
    try:
        for variant in (12, 6):
            N.check(N.lib().bsi_gemm_set_variant(variant))
            for rep in range(2):  # twice: a race would rarely repeat identically
                out = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
                a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K,
                               ldo=Nn, epilogue=epi)
                N.check(N.lib().bsi_gemm_bf16(C.byref(a), N.stream()))
                torch.cuda.synchronize()
                if rep:
                    assert torch.equal(out, outs[variant]), f"variant {variant} is not reproducible"
                outs[variant] = out
    finally:
        N.check(N.lib().bsi_gemm_set_variant(12))
    o12, o6 = outs[12].float(), outs[6].float()
    assert bool(torch.isfinite(o12).all())
    # the two schedules differ only in where the bias joins the fp32 sum: at most a bf16 ulp apart
    assert float(((o12 - o6).abs() / (o6.abs() + 0.05)).max()) < 1.6e-2
    rows = torch.randint(0, M, (192,), generator=gen)
    ref = A[rows].double() @ W.double().t() + bias.double()
    if epi_name == "gelu":
        ref = do.gelu_tanh(ref)
    assert rel_linf(o12[rows.to(DEV)].cpu(), ref) < 5e-3


@pytest.mark.parametrize("M,Nn,K", [(256, 1024, 4096), (300, 3072, 2048), (1024, 1024, 4096), (2048, 4096, 2048), (512, 768, 2048)])
def test_gemm_split_k_small_m(N, M, Nn, K):
    """bsi_gemm_bf16_ws: the split-K latency path for a few images per call against fp64 and against bsi_gemm_bf16, all three plain
    bf16 epilogues; with too small a workspace it must fall back to the ordinary kernel (same result as bsi_gemm_bf16)."""
    gen = torch.Generator().manual_seed(M + Nn + K)
    A = bf16r(torch.randn((M, K), generator=gen))
    W = bf16r(torch.randn((Nn, K), generator=gen) / math.sqrt(K))
    bias = torch.randn(Nn, generator=gen)
    ref = A.double() @ W.double().t() + bias.double()
    dA, dW, db = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(bias)
    lib = N.lib()
    need = lib.bsi_gemm_splitk_workspace_bytes(M, Nn, K)
    assert need > 0, "these shapes are meant to split"
    ws = empty(need, dtype=torch.uint8)
    for epi, fn in ((N.EPI_BIAS_BF16, lambda r: r), (N.EPI_BIAS_GELU_BF16, do.gelu_tanh), (N.EPI_BIAS_SILU_BF16, do.silu)):
        outs = []
        for mode in ("split", "small_ws", "plain"):
            out = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
            a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                           epilogue=epi)
            if mode == "split":
                N.check(lib.bsi_gemm_bf16_ws(C.byref(a), N.ptr(ws), need, N.stream()))
            elif mode == "small_ws":
                N.check(lib.bsi_gemm_bf16_ws(C.byref(a), N.ptr(ws), need // 2 - 1, N.stream()))
            else:
                N.check(lib.bsi_gemm_bf16(C.byref(a), N.stream()))
            outs.append(out.cpu().float())
        assert rel_linf(outs[0], fn(ref)) < 5e-3
        assert torch.equal(outs[1], outs[2])                                              # fallback = the ordinary kernel
        assert float(((outs[0] - outs[2]).abs() / (outs[2].abs() + 0.05)).max()) < 1.6e-2  # split vs ordinary: a bf16 ulp


@pytest.mark.parametrize("M,Nn,K,G,shared_a", [(512, 1024, 1024, 5, True), (512, 768, 256, 3, False), (8, 512, 1024, 4, False), (200, 320, 192, 2, True)])
def test_gemm_grouped_matches_one_launch_per_group(N, M, Nn, K, G, shared_a):
    """bsi_gemm_bf16_grouped: G problems of one shape (the per-sample adaLN matrices of the DiT blocks: shared or per-group A, per-group W,
    bias and output at uniform byte strides, the output strided inside a wider table) in one launch against fp64, bit-identical to one
    bsi_gemm_bf16 launch per group where that runs the same kernel (M > 128), and with every row independent of the batch it sits in
    (rows of an M = 8 call equal the same rows inside the M-row call)."""
    gen = torch.Generator().manual_seed(M + Nn + K + G)
    A = bf16r(torch.randn((1 if shared_a else G, M, K), generator=gen))
    W = bf16r(torch.randn((G, Nn, K), generator=gen) / math.sqrt(K))
    bias = torch.randn((G, Nn), generator=gen)
    dA, dW, db = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(bias)
    out = torch.full((M, G * Nn + 8), float("nan"), device=DEV)  # group g: columns g * Nn .. of a wider table (the adaLN chunk table)
    ld = G * Nn + 8
    a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=ld,
                   epilogue=N.EPI_BIAS_F32)
    N.check(N.lib().bsi_gemm_bf16_grouped(C.byref(a), G, 0 if shared_a else M * K * 2, Nn * K * 2, Nn * 4, Nn * 4, N.stream()))
    for g in range(G):
        ref = A[0 if shared_a else g].double() @ W[g].double().t() + bias[g].double()
        got = out[:, g * Nn:(g + 1) * Nn]
        assert rel_linf(got, ref) < 2e-5, (g, rel_linf(got, ref))
        if M > 128 and K >= 96:
            one = torch.empty((M, Nn), device=DEV)
            b = N.GemmArgs(A=dA[0 if shared_a else g].data_ptr(), W=dW[g].data_ptr(), bias=db[g].data_ptr(), out=one.data_ptr(), M=M, N=Nn, K=K,
                           lda=K, ldw=K, ldo=Nn, epilogue=N.EPI_BIAS_F32)
            N.check(N.lib().bsi_gemm_bf16(C.byref(b), N.stream()))
            assert torch.equal(one, got.contiguous())
    assert torch.isnan(out[:, G * Nn:]).all()
    if M >= 16:  # batch independence: the first 8 rows as a call of their own
        small = torch.empty((8, ld), device=DEV)
        a8 = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=small.data_ptr(), M=8, N=Nn, K=K, lda=K, ldw=K, ldo=ld,
                        epilogue=N.EPI_BIAS_F32)
        N.check(N.lib().bsi_gemm_bf16_grouped(C.byref(a8), G, 0 if shared_a else M * K * 2, Nn * K * 2, Nn * 4, Nn * 4, N.stream()))
        assert torch.equal(small[:, :G * Nn], out[:8, :G * Nn])
    assert N.lib().bsi_gemm_bf16_grouped(C.byref(a), G, 8, Nn * K * 2, Nn * 4, Nn * 4, N.stream()) != 0  # stride not a multiple of 16


@pytest.mark.parametrize("Nn,K", [(1024, 1024), (6144, 1024), (1024, 6144)])
def test_gemm_split_k_fp32_per_sample_rows(N, Nn, K):
    """The per-sample GEMMs of the train step (adaLN MLP, dit.py:77-81: M = images) with the fp32 epilogue through bsi_gemm_bf16_ws:
    against fp64, and -- the slice count depends on K only -- the rows of a batch equal the rows of its first half BIT FOR BIT
    (a batch and its shards must see the same conditioning table)."""
    gen = torch.Generator().manual_seed(Nn + K)
    M = 96
    A = bf16r(torch.randn((M, K), generator=gen))
    W = bf16r(torch.randn((Nn, K), generator=gen) / math.sqrt(K))
    bias = torch.randn(Nn, generator=gen)
    ref = A.double() @ W.double().t() + bias.double()
    dA, dW, db = dev(A.to(torch.bfloat16)), dev(W.to(torch.bfloat16)), dev(bias)
    lib = N.lib()
    need = lib.bsi_gemm_splitk_f32_workspace_bytes(M, Nn, K)
    assert need > 0, "these shapes are meant to split"
    ws = empty(need, dtype=torch.uint8)
    outs = {}
    for rows in (M, M // 2):
        out = torch.full((rows, Nn), float("nan"), dtype=torch.float32, device=DEV)
        a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=out.data_ptr(), M=rows, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                       epilogue=N.EPI_BIAS_F32)
        N.check(lib.bsi_gemm_bf16_ws(C.byref(a), N.ptr(ws), need, N.stream()))
        outs[rows] = out.cpu()
    assert rel_linf(outs[M], ref) < 2e-5                      # fp32 accumulation of bf16 products
    assert torch.equal(outs[M][: M // 2], outs[M // 2])       # independent of how many rows were computed with it
    plain = torch.empty((M, Nn), dtype=torch.float32, device=DEV)
    a = N.GemmArgs(A=dA.data_ptr(), W=dW.data_ptr(), bias=db.data_ptr(), out=plain.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                   epilogue=N.EPI_BIAS_F32)
    N.check(lib.bsi_gemm_bf16(C.byref(a), N.stream()))
    assert rel_linf(outs[M], plain.cpu()) < 2e-5              # the unsplit kernel: same value up to the summation order


def test_cu_reserve_changes_grids_not_results(N):
    """bsi_set_cu_reserve (the data-parallel step's CU budget): the persistent kernels launch on fewer CUs and produce the same bits --
    the forward GEMM (K = 64 ring), the weight-gradient GEMM (split count follows the budget), attention forward and backward."""
    lib = N.lib()
    full = lib.bsi_compute_cus()
    assert lib.bsi_set_cu_reserve(12) != 0 and b"multiple of 8" in lib.bsi_last_error()   # one share per XCD
    assert lib.bsi_set_cu_reserve(72) != 0
    gen = torch.Generator().manual_seed(3)
    M, Nn, K = 2048, 1024, 1024
    A = dev(torch.randn((M, K), generator=gen).to(torch.bfloat16))
    W = dev((torch.randn((Nn, K), generator=gen) / 32).to(torch.bfloat16))
    bias = dev(torch.randn(Nn, generator=gen))
    T, H, dh, B = 256, 4, 64, 70
    qkv = dev(torch.randn((B, T, 3 * H * dh), generator=gen).to(torch.bfloat16))
    dout = dev(torch.randn((B, T, H * dh), generator=gen).to(torch.bfloat16))

    def run():
        out = torch.empty((M, Nn), dtype=torch.bfloat16, device=DEV)
        a = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                       epilogue=N.EPI_BIAS_GELU_BF16)
        N.check(lib.bsi_gemm_bf16(C.byref(a), N.stream()))
        dw = torch.empty((Nn, K), dtype=torch.float32, device=DEV)
        ws = empty(lib.bsi_gemm_tn_workspace_bytes(M, Nn, K), dtype=torch.uint8)
        N.check(lib.bsi_gemm_tn_bf16(N.ptr(out), Nn, N.ptr(A), K, M, Nn, K, N.ptr(dw), K, 0, N.ptr(ws), N.stream()))
        ao = torch.empty((B, T, H * dh), dtype=torch.bfloat16, device=DEV)
        lse = torch.empty((B, H, T), device=DEV)
        N.check(lib.bsi_attention_fwd_lse(N.ptr(qkv), 3 * H * dh, B, T, H, dh, N.ptr(ao), H * dh, N.ptr(lse), N.stream()))
        dq = torch.empty_like(qkv)
        N.check(lib.bsi_attention_bwd(N.ptr(qkv), 3 * H * dh, N.ptr(ao), N.ptr(dout), H * dh, N.ptr(lse), B, T, H, dh, N.ptr(dq), 3 * H * dh,
                                      N.stream()))
        torch.cuda.synchronize()
        return out.cpu(), dw.cpu(), ao.cpu(), dq.cpu()

    base = run()
    try:
        N.check(lib.bsi_set_cu_reserve(32))
        assert lib.bsi_compute_cus() == full - 32
        held = run()
    finally:
        N.check(lib.bsi_set_cu_reserve(0))
    assert lib.bsi_compute_cus() == full
    assert torch.equal(base[0], held[0]) and torch.equal(base[2], held[2]) and torch.equal(base[3], held[3])
    # the weight gradient's split count follows the CU count: a different (still fixed) summation order
    assert rel_linf(held[1], base[1]) < 1e-5


@pytest.mark.parametrize("M,Nn,K", [(16384, 4096, 1024), (16384, 1024, 4096), (16384 + 77, 3072, 1024), (32768, 1024, 512), (70000, 1280, 1152)])
def test_gemm_tile_queue_is_bit_identical_to_the_static_schedule(N, M, Nn, K):
    """bsi_set_tile_queue(1): workgroups of the persistent K = 64 GEMM draw tile tickets from per-XCD counters (and from the other
    XCDs' once theirs is dry) instead of a static share.  Same tiles, same arithmetic per tile: the outputs must be the static
    schedule's bits for every epilogue that kernel takes -- over repeated launches (the last workgroup resets the counters), with a
    CU reserve in force (the queue's grid ignores it), with ragged M, and on a second stream (its own control block)."""
    lib = N.lib()
    gen = torch.Generator().manual_seed(M + Nn + K)
    A = dev(torch.randn((M, K), generator=gen).to(torch.bfloat16))
    W = dev((torch.randn((Nn, K), generator=gen) / 32).to(torch.bfloat16))
    bias = dev(torch.randn(Nn, generator=gen))
    aux = dev(torch.randn((M, Nn), generator=gen).to(torch.bfloat16))

    def run(epi, stream=None):
        out = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV)
        out2 = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device=DEV) if epi == N.EPI_BIAS_GELU_DUAL else None
        a = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                       epilogue=epi, aux=aux.data_ptr() if epi == N.EPI_MUL_GELUGRAD_BF16 else None,
                       out2=out2.data_ptr() if out2 is not None else None)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else N.stream()
        N.check(lib.bsi_gemm_bf16(C.byref(a), st))
        return out, out2

    epis = [N.EPI_BIAS_BF16, N.EPI_BIAS_GELU_BF16, N.EPI_BIAS_GELU_DUAL, N.EPI_MUL_GELUGRAD_BF16]
    N.check(lib.bsi_set_tile_queue(0))
    base = {e: run(e) for e in epis}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    try:
        N.check(lib.bsi_set_tile_queue(1))
        assert lib.bsi_set_tile_queue(2) != 0
        for rep in range(3):
            if rep == 2:
                N.check(lib.bsi_set_cu_reserve(32))
            for e in epis:
                got = run(e)
                assert torch.equal(got[0], base[e][0]) and (got[1] is None or torch.equal(got[1], base[e][1])), (rep, e)
        N.check(lib.bsi_set_cu_reserve(0))
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a_, _ = run(N.EPI_BIAS_GELU_BF16, side)
        b_, _ = run(N.EPI_BIAS_BF16)           # concurrently on the current stream: each stream has its own counters
        side.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(a_, base[N.EPI_BIAS_GELU_BF16][0]) and torch.equal(b_, base[N.EPI_BIAS_BF16][0])
    finally:
        N.check(lib.bsi_set_cu_reserve(0))
        N.check(lib.bsi_set_tile_queue(0))


@pytest.mark.parametrize("p", [0.05, 0.3])
def test_attention_dropout_words_equal_the_exported_mask(N, p):
    """The lane-mask words of the 256-token attention kernels (bsi_attention_dropout_words) against bsi_dropout_mask for the same
    (p, seed, site): word (pair, qb, kt, r), bit 16 g + c <-> element (row = pair * 256 + 16 qb + c, column = 16 kt + 4 g + r)."""
    lib = N.lib()
    pairs, seed, site = 6, 0x1234567890ABCDEF, 4
    keep = torch.empty(pairs * 256 * 256, dtype=torch.uint8, device=DEV)
    N.check(lib.bsi_dropout_mask(p, seed, site, pairs * 256, 256, N.ptr(keep), N.stream()))
    words = torch.empty(pairs * 16 * 64, dtype=torch.int64, device=DEV)
    N.check(lib.bsi_attention_dropout_words(p, seed, site, pairs, N.ptr(words), N.stream()))
    torch.cuda.synchronize()
    w = words.cpu().reshape(pairs, 16, 16, 4)
    bits = ((w.unsqueeze(-1) >> torch.arange(64)) & 1).reshape(pairs, 16, 16, 4, 4, 16)   # pair, qb, kt, r, g, c
    m = bits.permute(0, 1, 5, 2, 4, 3).reshape(pairs, 256, 256)                             # query 16 qb + c, key 16 kt + 4 g + r
    k = keep.cpu().reshape(pairs, 256, 256).to(torch.int64)
    assert torch.equal(m, k)
    assert abs(float(k.float().mean()) - (1 - p)) < 5e-3


def test_gemm_rejects_bad_shapes(N):
    a = N.GemmArgs(A=1, W=1, out=1, M=4, N=24, K=64, lda=64, ldw=64, ldo=24, epilogue=0)
    assert N.lib().bsi_gemm_bf16(C.byref(a), None) == -1
    assert b"multiple of 16" in N.lib().bsi_last_error()
    a = N.GemmArgs(A=1, W=1, out=1, M=4, N=32, K=100, lda=104, ldw=104, ldo=32, epilogue=0)
    assert N.lib().bsi_gemm_bf16(C.byref(a), None) == -1


@pytest.mark.parametrize("B,tokens,heads,dh", [(2, 64, 2, 64), (3, 256, 4, 64), (2, 256, 16, 64), (1, 1024, 1, 128),
                                                (2, 64, 1, 128), (1, 512, 2, 64), (41, 256, 16, 64)])
def test_attention(N, B, tokens, heads, dh):
    gen = torch.Generator().manual_seed(B + tokens + heads)
    d = heads * dh
    qkv = bf16r(torch.randn((B, tokens, 3, heads, dh), generator=gen) * 1.5)
    out = empty(B, tokens, d, dtype=torch.bfloat16)
    N.check(N.lib().bsi_attention_fwd(N.ptr(dev(qkv.to(torch.bfloat16))), 3 * d, B, tokens, heads, dh, N.ptr(out), d,
                                      N.stream()))
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).double() for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, tokens, d)
    # P is rounded to bf16 before the PV product (2^-9 relative per term, averaged) and O is stored as bf16
    assert rel_linf(out.cpu().float(), ref) < 1e-2, rel_linf(out.cpu().float(), ref)
    assert float((out.cpu().double() - ref).abs().mean() / ref.abs().mean()) < 4e-3
    if dh == 64:  # the variant that also saves the log-sum-exp (training forward); 41 x 16 pairs: 2-3 pairs per persistent workgroup
        out2, lse = empty(B, tokens, d, dtype=torch.bfloat16), empty(B, heads, tokens)
        N.check(N.lib().bsi_attention_fwd_lse(N.ptr(dev(qkv.to(torch.bfloat16))), 3 * d, B, tokens, heads, dh, N.ptr(out2), d,
                                              N.ptr(lse), N.stream()))
        assert torch.equal(out2, out)
        ref_lse = torch.logsumexp(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
        assert float((lse.cpu().double() - ref_lse).abs().max()) < 2e-3


@pytest.mark.parametrize("M,d,tokens,mod_rows", [(128, 128, 64, 2), (512, 1024, 256, 1), (512, 1024, 256, 2),
                                                  (77, 256, 1, 77)])
def test_ln_modulate(N, M, d, tokens, mod_rows):
    gen = torch.Generator().manual_seed(M + d)
    x = torch.randn((M, d), generator=gen) * 2 + 0.3
    mod = torch.randn((mod_rows, 6 * d), generator=gen) * 0.2
    out = empty(M, d, dtype=torch.bfloat16)
    dmod = dev(mod)
    N.check(N.lib().bsi_ln_modulate(N.ptr(dev(x)), M, d, 1e-5, dmod.data_ptr() + 4 * 3 * d, dmod.data_ptr() + 4 * 4 * d,
                                    mod_rows, 6 * d, tokens, None, None, N.ptr(out), N.stream()))
    rows = (torch.arange(M) // tokens) % mod_rows
    ref = torch.addcmul(mod[rows, 3 * d:4 * d], mod[rows, 4 * d:5 * d] + 1, do.layer_norm(x))
    # fp32 statistics, one bf16 rounding of the result
    assert torch.equal(out.cpu(), ref.to(torch.bfloat16)) or rel_linf(out.cpu().float(), ref) < 4e-3
    assert float((out.cpu().float() - ref).abs().max()) <= float(ref.abs().max()) * 2 ** -8
    w, b = torch.randn(d, generator=gen), torch.randn(d, generator=gen)
    N.check(N.lib().bsi_ln_modulate(N.ptr(dev(x)), M, d, 1e-5, None, None, 0, 0, 1, N.ptr(dev(w)), N.ptr(dev(b)),
                                    N.ptr(out), N.stream()))
    ref = do.layer_norm(x, 1e-5, w, b)
    assert float((out.cpu().float() - ref).abs().max()) <= float(ref.abs().max()) * 2 ** -8


@pytest.mark.parametrize("M,d,tokens,mod_rows", [(128, 128, 64, 2), (512, 1024, 256, 2), (256, 1024, 256, 1)])
def test_resid_ln_modulate(N, M, d, tokens, mod_rows):
    """Fused gated residual update + LayerNorm + modulate (dit.py:93-102 followed by :96 of the next branch)."""
    gen = torch.Generator().manual_seed(M * 3 + d)
    x = torch.randn((M, d), generator=gen) * 2
    delta = bf16r(torch.randn((M, d), generator=gen))
    mod = torch.randn((mod_rows, 6 * d), generator=gen) * 0.3
    rows = (torch.arange(M) // tokens) % mod_rows
    x_ref = torch.addcmul(x, mod[rows, 2 * d:3 * d], delta)
    y_ref = torch.addcmul(mod[rows, 3 * d:4 * d], mod[rows, 4 * d:5 * d] + 1, do.layer_norm(x_ref))
    dx, dmod = dev(x.clone()), dev(mod)
    buf = dev(delta.to(torch.bfloat16))  # delta aliases the output buffer, as in the engine
    N.check(N.lib().bsi_resid_ln_modulate(N.ptr(dx), M, d, 1e-5, N.ptr(buf), dmod.data_ptr() + 4 * 2 * d,
                                          dmod.data_ptr() + 4 * 3 * d, dmod.data_ptr() + 4 * 4 * d, mod_rows, 6 * d,
                                          tokens, None, None, N.ptr(buf), N.stream()))
    assert torch.equal(dx.cpu(), x_ref)  # fp32 fma, exact
    assert float((buf.cpu().float() - y_ref).abs().max()) <= float(y_ref.abs().max()) * 2 ** -8
    # pure residual update (no norm)
    dx2 = dev(x.clone())
    dd = dev(delta.to(torch.bfloat16))
    N.check(N.lib().bsi_resid_ln_modulate(N.ptr(dx2), M, d, 1e-5, N.ptr(dd), dmod.data_ptr() + 4 * 2 * d, None, None,
                                          mod_rows, 6 * d, tokens, None, None, None, N.stream()))
    assert torch.equal(dx2.cpu(), x_ref)


@pytest.mark.parametrize("M,d,tokens,mod_rows", [(128, 128, 64, 2), (512, 1024, 256, 2), (300, 2048, 100, 3)])
def test_resid2_ln_modulate_lazy_store_is_bit_identical(N, M, d, tokens, mod_rows):
    """The inference engine's lazy form of two consecutive updates (dit.py:93-102): the first pass applies the attention update in
    registers only (write_x = 0), the second names it as (delta0, gate0) and stores the row once.  Rows and both normalised
    outputs must equal the two stored passes of bsi_resid_ln_modulate bit for bit; the second delta aliases the output buffer as
    in the engine."""
    gen = torch.Generator().manual_seed(M * 5 + d)
    x = torch.randn((M, d), generator=gen) * 2
    da = bf16r(torch.randn((M, d), generator=gen)).to(torch.bfloat16)
    dm = bf16r(torch.randn((M, d), generator=gen)).to(torch.bfloat16)
    mod = torch.randn((mod_rows, 6 * d), generator=gen) * 0.3
    dmod = dev(mod)
    g = lambda j: dmod.data_ptr() + 4 * j * d  # noqa: E731  chunk j of the modulation table
    L = N.lib()
    # eager: two stored passes
    xe, b1, b2 = dev(x.clone()), dev(da.clone()), dev(dm.clone())
    o1 = empty(M, d, dtype=torch.bfloat16)
    N.check(L.bsi_resid_ln_modulate(N.ptr(xe), M, d, 1e-5, N.ptr(b1), g(2), g(3), g(4), mod_rows, 6 * d, tokens, None, None,
                                    N.ptr(o1), N.stream()))
    N.check(L.bsi_resid_ln_modulate(N.ptr(xe), M, d, 1e-5, N.ptr(b2), g(5), g(0), g(1), mod_rows, 6 * d, tokens, None, None,
                                    N.ptr(b2), N.stream()))
    # lazy: the row is stored once
    xl, c1, c2 = dev(x.clone()), dev(da.clone()), dev(dm.clone())
    p1 = empty(M, d, dtype=torch.bfloat16)
    N.check(L.bsi_resid2_ln_modulate(N.ptr(xl), M, d, 1e-5, None, None, N.ptr(c1), g(2), 0, g(3), g(4), mod_rows, 6 * d, tokens,
                                     N.ptr(p1), N.stream()))
    assert torch.equal(xl.cpu(), x)  # untouched by the register-only pass
    assert torch.equal(p1.cpu(), o1.cpu())
    N.check(L.bsi_resid2_ln_modulate(N.ptr(xl), M, d, 1e-5, N.ptr(c1), g(2), N.ptr(c2), g(5), 1, g(0), g(1), mod_rows, 6 * d, tokens,
                                     N.ptr(c2), N.stream()))
    assert torch.equal(xl.cpu(), xe.cpu()) and torch.equal(c2.cpu(), b2.cpu())
    rows = (torch.arange(M) // tokens) % mod_rows
    x_ref = torch.addcmul(torch.addcmul(x, mod[rows, 2 * d:3 * d], da.float()), mod[rows, 5 * d:6 * d], dm.float())
    assert torch.equal(xl.cpu(), x_ref)  # two fp32 fmas, exact
    # argument checks: an older update without a newer one, a register-only pass without output
    assert L.bsi_resid2_ln_modulate(N.ptr(xl), M, d, 1e-5, N.ptr(c1), g(2), None, None, 1, g(0), g(1), mod_rows, 6 * d, tokens,
                                    N.ptr(c2), N.stream()) != 0
    assert L.bsi_resid2_ln_modulate(N.ptr(xl), M, d, 1e-5, None, None, N.ptr(c1), g(2), 0, g(0), g(1), mod_rows, 6 * d, tokens,
                                    None, N.stream()) != 0


@pytest.mark.parametrize("M,Nn,K", [(512, 256, 256), (4096, 1024, 1024), (16384, 3072, 1024), (8192, 1024, 4096),
                                     (1000 * 32, 128, 384), (544, 512, 128), (4, 768, 128), (300, 64, 64)])
def test_gemm_tn_and_colsum(N, M, Nn, K):
    """dW = dY^T X and db = colsum(dY) (backward of nn.Linear) vs fp64 on the same bf16 operands."""
    gen = torch.Generator().manual_seed(M + Nn + K)
    dY = bf16r(torch.randn((M, Nn), generator=gen))
    X = bf16r(torch.randn((M, K), generator=gen))
    ref = dY.double().t() @ X.double()
    out = empty(Nn, K)
    ws = torch.empty(N.lib().bsi_gemm_tn_workspace_bytes(M, Nn, K), dtype=torch.uint8, device=DEV)
    dP, dQ = dev(dY.to(torch.bfloat16)), dev(X.to(torch.bfloat16))
    N.check(N.lib().bsi_gemm_tn_bf16(N.ptr(dP), Nn, N.ptr(dQ), K, M, Nn, K, N.ptr(out), K, 0, N.ptr(ws), N.stream()))
    # exact bf16 products, fp32 accumulation over M terms: error ~ sqrt(M) * 2^-24 of the row scale
    assert rel_linf(out, ref) < 3e-5, rel_linf(out, ref)
    N.check(N.lib().bsi_gemm_tn_bf16(N.ptr(dP), Nn, N.ptr(dQ), K, M, Nn, K, N.ptr(out), K, 1, N.ptr(ws), N.stream()))
    assert rel_linf(out, 2 * ref) < 3e-5  # accumulate
    cs = empty(Nn)
    ws2 = torch.empty(N.lib().bsi_colsum_workspace_bytes(Nn), dtype=torch.uint8, device=DEV)
    N.check(N.lib().bsi_colsum_bf16(N.ptr(dP), Nn, M, Nn, N.ptr(cs), 0, N.ptr(ws2), N.stream()))
    assert rel_linf(cs, dY.double().sum(0)) < 1e-5
    # fused: weight and bias gradient from one launch (+ accumulate)
    out2, cs2 = empty(Nn, K), empty(Nn)
    for acc_flag, mult in ((0, 1), (1, 2)):
        N.check(N.lib().bsi_gemm_tn_bias_bf16(N.ptr(dP), Nn, N.ptr(dQ), K, M, Nn, K, N.ptr(out2), K, N.ptr(cs2), acc_flag, N.ptr(ws),
                                              N.stream()))
        assert rel_linf(out2, mult * ref) < 3e-5
        assert rel_linf(cs2, mult * dY.double().sum(0)) < 1e-5, rel_linf(cs2, mult * dY.double().sum(0))


@pytest.mark.parametrize("M,N1,N2,K,ld1", [(16384, 3072, 1024, 1024, 3072), (4096 + 40, 768, 200, 256, 1024), (700, 256, 256, 320, 256)])
def test_gemm_tn_pair_matches_two_launches(N, M, N1, N2, K, ld1):
    """bsi_gemm_tn_pair_bf16: two weight gradients of one token count in one launch (the DiT block's qkv + out-projection gradients:
    64 output tiles together) against fp64 on the same bf16 operands and against the two single launches (another split of the token
    range: equal to fp32 summation order); ragged token counts / second N / padded leading dimension; run twice (reproducible)."""
    gen = torch.Generator().manual_seed(M + N1 + N2)
    dY1 = bf16r(torch.randn((M, ld1), generator=gen))
    dY2 = bf16r(torch.randn((M, N2), generator=gen))
    X1, X2 = bf16r(torch.randn((M, K), generator=gen)), bf16r(torch.randn((M, K), generator=gen))
    r1, r2 = dY1[:, :N1].double().t() @ X1.double(), dY2.double().t() @ X2.double()
    d1, d2, q1, q2 = (dev(t.to(torch.bfloat16)) for t in (dY1, dY2, X1, X2))
    ws = torch.empty(N.lib().bsi_gemm_tn_workspace_bytes(M, N1 + (N2 + 255) // 256 * 256, K), dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        o1 = torch.full((N1, K), float("nan"), device=DEV)
        o2 = torch.full((N2, K), float("nan"), device=DEV)
        N.check(N.lib().bsi_gemm_tn_pair_bf16(N.ptr(d1), ld1, N.ptr(q1), N1, N.ptr(o1), N.ptr(d2), N2, N.ptr(q2), N2, N.ptr(o2), K, M, K,
                                              N.ptr(ws), N.stream()))
        torch.cuda.synchronize()
        outs.append((o1, o2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel_linf(outs[0][0], r1) < 3e-5 and rel_linf(outs[0][1], r2) < 3e-5
    s1, s2 = empty(N1, K), empty(N2, K)
    N.check(N.lib().bsi_gemm_tn_bf16(N.ptr(d1), ld1, N.ptr(q1), K, M, N1, K, N.ptr(s1), K, 0, N.ptr(ws), N.stream()))
    N.check(N.lib().bsi_gemm_tn_bf16(N.ptr(d2), N2, N.ptr(q2), K, M, N2, K, N.ptr(s2), K, 0, N.ptr(ws), N.stream()))
    assert rel_linf(outs[0][0], s1.cpu().double()) < 2e-5 and rel_linf(outs[0][1], s2.cpu().double()) < 2e-5


# ----------------------------------------------------------------------------------------------
# backward kernels (checked against torch autograd of the oracle expressions in fp64)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,tokens,heads,dh,long", [(2, 64, 2, 64, 0), (2, 256, 3, 64, 0), (20, 256, 16, 64, 0), (70, 256, 16, 64, 0), (1, 128, 1, 64, 0), (2, 1024, 1, 128, 1),
                                                    (2, 192, 2, 128, 1), (1, 320, 2, 64, 1), (3, 64, 1, 64, 1)])
def test_attention_backward(N, B, tokens, heads, dh, long):
    d = heads * dh
    gen = torch.Generator().manual_seed(B * 5 + tokens + heads)
    qkv = bf16r(torch.randn((B, tokens, 3, heads, dh), generator=gen) * 1.2)
    dout = bf16r(torch.randn((B, tokens, d), generator=gen))
    dq_ = dev(qkv.to(torch.bfloat16))
    out = empty(B, tokens, d, dtype=torch.bfloat16)
    lse = empty(B, heads, tokens)
    N.check(N.lib().bsi_attention_fwd_lse(N.ptr(dq_), 3 * d, B, tokens, heads, dh, N.ptr(out), d, N.ptr(lse), N.stream()))
    x = qkv.double().requires_grad_(True)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    ref_o = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, tokens, d)
    assert rel_linf(lse, torch.logsumexp(sc, -1).detach()) < 1e-5
    ref_o.backward(dout.double())
    dqkv = empty(B, tokens, 3 * d, dtype=torch.bfloat16)
    fn = N.lib().bsi_attention_bwd_long if long else N.lib().bsi_attention_bwd
    N.check(fn(N.ptr(dq_), 3 * d, N.ptr(out), N.ptr(dev(dout.to(torch.bfloat16))), d, N.ptr(lse), B, tokens, heads, dh, N.ptr(dqkv),
               3 * d, N.stream()))
    got = dqkv.cpu().float().reshape(B, tokens, 3, heads, dh)
    # P and dS are rounded to bf16 before the second products, O and the outputs are bf16: 1e-2 relative to the max
    for i, nm in enumerate("qkv"):
        assert rel_linf(got[:, :, i], x.grad[:, :, i]) < 1.5e-2, (nm, rel_linf(got[:, :, i], x.grad[:, :, i]))
        assert float((got[:, :, i].double() - x.grad[:, :, i]).abs().mean() / x.grad[:, :, i].abs().mean()) < 6e-3


@pytest.mark.parametrize("B,heads,p,words", [(3, 2, 0.3, True), (3, 2, 0.3, False), (70, 16, 0.1, True)])
def test_attention_dropout_forward_backward(N, B, heads, p, words):
    """bsi_attention_fwd_dropout / bsi_attention_bwd_dropout (dit.py:43-44 in training) on the DiT geometry against fp64 autograd of
    softmax(q k^T / sqrt(dh)) * keep / (1 - p) @ v with the keep flags bsi_dropout_mask exports for the same (seed, site): with the mask
    words on a tape buffer (the persistent forward and the single-sweep backward: up to 1120 (image, head) pairs, 4-5 per compute
    unit, so the rolling refill of the tiles, statistics and words of the NEXT pair is exercised through several hand-overs) and
    without (both sides evaluate the hash).  Run twice: the same bits."""
    tokens, dh = 256, 64
    d = heads * dh
    seed, site = 20240917, 6
    gen = torch.Generator().manual_seed(B + heads)
    qkv = bf16r(torch.randn((B, tokens, 3, heads, dh), generator=gen) * 1.1)
    dout = bf16r(torch.randn((B, tokens, d), generator=gen))
    dq_, dd_ = dev(qkv.to(torch.bfloat16)), dev(dout.to(torch.bfloat16))
    keep = torch.empty(B * heads * tokens * tokens, dtype=torch.uint8, device=DEV)
    N.check(N.lib().bsi_dropout_mask(p, seed, site, B * heads * tokens, tokens, N.ptr(keep), N.stream()))
    keep = keep.reshape(B, heads, tokens, tokens).cpu().double()
    pq = round(p * 65536) / 65536  # the kernels apply the probability rounded to 2^-16 and scale the survivors by its complement
    out = empty(B, tokens, d, dtype=torch.bfloat16)
    lse = empty(B, heads, tokens)
    dqkv = torch.full((B, tokens, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
    mw = torch.zeros(B * heads * 8192, dtype=torch.uint8, device=DEV) if words else None
    def run():
        N.check(N.lib().bsi_attention_fwd_dropout(N.ptr(dq_), 3 * d, B, tokens, heads, dh, N.ptr(out), d, N.ptr(lse), p, seed, site,
                                                  N.ptr(mw) if words else None, N.stream()))
        N.check(N.lib().bsi_attention_bwd_dropout(N.ptr(dq_), 3 * d, N.ptr(out), N.ptr(dd_), d, N.ptr(lse), B, tokens, heads, dh, N.ptr(dqkv),
                                                  3 * d, p, seed, site, N.ptr(mw) if words else None, N.stream()))
    run()
    x = qkv.double().requires_grad_(True)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    ref_o = ((torch.softmax(sc, -1) * keep / (1 - pq)) @ v).permute(0, 2, 1, 3).reshape(B, tokens, d)
    assert rel_linf(lse, torch.logsumexp(sc, -1).detach()) < 1e-5
    assert rel_linf(out.cpu().float(), ref_o.detach()) < 1.5e-2
    ref_o.backward(dout.double())
    got = dqkv.cpu().float().reshape(B, tokens, 3, heads, dh)
    assert torch.isfinite(got).all()
    for i, nm in enumerate("qkv"):
        assert rel_linf(got[:, :, i], x.grad[:, :, i]) < 1.5e-2, (nm, rel_linf(got[:, :, i], x.grad[:, :, i]))
        assert float((got[:, :, i].double() - x.grad[:, :, i]).abs().mean() / x.grad[:, :, i].abs().mean()) < 6e-3
        # per (image, head): a hand-over that lost one pair's statistics or tiles shows here, not in the maximum over the batch
        err = (got[:, :, i].double() - x.grad[:, :, i]).abs().amax(dim=(1, 3)) / x.grad[:, :, i].abs().amax(dim=(1, 3))
        assert float(err.max()) < 3e-2, (nm, int(err.argmax()), float(err.max()))
    first_o, first_g = out.clone(), dqkv.clone()
    run()
    assert torch.equal(first_o.view(torch.int16), out.view(torch.int16)) and torch.equal(first_g.view(torch.int16), dqkv.view(torch.int16))


@pytest.mark.parametrize("B,p", [(70, 0.1), (70, 0.0), (512, 0.1)])
def test_attention_backward_schedules_agree_bit_for_bit(N, B, p):
    """The single-sweep attention backward with its two wave groups half a trip apart (bsi_set_attention_bwd_skew(1), the default) and
    in lock step (0) runs the same arithmetic in the same order: dQ, dK and dV agree bit for bit -- at 1120 pairs (4-5 per compute
    unit) and at the training batch (8192 pairs, 32 per compute unit), so a hand-over between slots that read a tile, an exchange
    buffer, a statistic or a mask word too early or too late in either schedule shows as a difference."""
    tokens, heads, dh = 256, 16, 64
    d = heads * dh
    seed, site = 77, 3
    gen = torch.Generator(device=DEV).manual_seed(B)
    qkv = (torch.randn((B, tokens, 3 * d), device=DEV, generator=gen) * 1.1).to(torch.bfloat16)
    dout = torch.randn((B, tokens, d), device=DEV, generator=gen).to(torch.bfloat16)
    out = empty(B, tokens, d, dtype=torch.bfloat16)
    lse = empty(B, heads, tokens)
    mw = torch.zeros(B * heads * 8192, dtype=torch.uint8, device=DEV) if p else None
    N.check(N.lib().bsi_attention_fwd_dropout(N.ptr(qkv), 3 * d, B, tokens, heads, dh, N.ptr(out), d, N.ptr(lse), p, seed, site,
                                              N.ptr(mw) if p else None, N.stream()))
    got = {}
    prev = N.lib().bsi_set_attention_bwd_skew(1)
    try:
        for on in (1, 0, 1):
            N.lib().bsi_set_attention_bwd_skew(on)
            dqkv = torch.full((B, tokens, 3 * d), float("nan"), dtype=torch.bfloat16, device=DEV)
            N.check(N.lib().bsi_attention_bwd_dropout(N.ptr(qkv), 3 * d, N.ptr(out), N.ptr(dout), d, N.ptr(lse), B, tokens, heads, dh, N.ptr(dqkv),
                                                      3 * d, p, seed, site, N.ptr(mw) if p else None, N.stream()))
            torch.cuda.synchronize()
            assert bool(torch.isfinite(dqkv.float()).all())
            if on in got:
                assert torch.equal(got[on].view(torch.int16), dqkv.view(torch.int16))
            got[on] = dqkv
    finally:
        N.lib().bsi_set_attention_bwd_skew(prev)
    diff = (got[0].view(torch.int16) != got[1].view(torch.int16))
    assert not bool(diff.any()), (int(diff.sum()), torch.nonzero(diff)[:4].tolist())


def test_copy_batch(N):
    """bsi_copy_batch_f32: many (src, dst, len) jobs in one launch -- lengths below, at and above a tile, not multiples of four,
    sources and destinations at byte offsets that are not multiples of 16; bytes outside the destinations stay untouched."""
    import ctypes as C
    gen = torch.Generator().manual_seed(3)
    lens = [1, 3, 256, 2047, 2048, 2049, 131072, 5, 70001]
    src_buf = dev(torch.randn(sum(lens) + 64, generator=gen))
    dst_buf = torch.full((sum(lens) + 4 * len(lens) + 64,), 7.0, device=DEV)
    descs, tiles, so, do = (N.CopyDesc * len(lens))(), 0, 1, 3  # element offsets 1 and 3: 4- and 12-byte misalignment
    jobs = []
    for i, n in enumerate(lens):
        s_, d_ = src_buf[so:so + n], dst_buf[do:do + n]
        descs[i].src, descs[i].dst, descs[i].len, descs[i].tile0 = s_.data_ptr(), d_.data_ptr(), n, tiles
        tiles += N.lib().bsi_copy_batch_tiles(n)
        jobs.append((so, do, n))
        so += n
        do += n + 4 if i % 2 else n + 1  # gaps, and alignments that differ between source and destination
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(DEV)
    N.check(N.lib().bsi_copy_batch_f32(N.ptr(table), len(lens), tiles, N.stream()))
    torch.cuda.synchronize()
    expect = torch.full_like(dst_buf, 7.0)
    for so, do, n in jobs:
        expect[do:do + n] = src_buf[so:so + n]
    assert torch.equal(dst_buf, expect)
    N.check(N.lib().bsi_copy_batch_f32(None, 0, 0, N.stream()))  # nothing to do


@pytest.mark.parametrize("Bs,tokens,d", [(3, 64, 128), (2, 256, 1024)])
def test_gate_and_ln_backward(N, Bs, tokens, d):
    gen = torch.Generator().manual_seed(Bs + tokens + d)
    M = Bs * tokens
    x1 = torch.randn((M, d), generator=gen) * 2
    delta = bf16r(torch.randn((M, d), generator=gen))
    mod = torch.randn((Bs, 6 * d), generator=gen) * 0.3
    dX = torch.randn((M, d), generator=gen)
    rows = torch.arange(M) // tokens
    gate = mod[:, 2 * d:3 * d]
    x2 = torch.addcmul(x1, gate[rows], delta)
    # --- gate backward
    dx2, dmod = dev(x2.clone()), dev(torch.zeros_like(mod))
    dd = empty(M, d, dtype=torch.bfloat16)
    dmod_in = dev(mod)
    N.check(N.lib().bsi_gate_bwd(N.ptr(dev(dX)), N.ptr(dev(delta.to(torch.bfloat16))), N.ptr(dx2), dmod_in.data_ptr() + 8 * d,
                                 6 * d, dmod.data_ptr() + 8 * d, 6 * d, M, d, tokens, N.ptr(dd), N.stream()))
    assert rel_linf(dx2, x1) < 1e-6                     # residual stream rewound to x1
    assert rel_linf(dd.cpu().float(), gate[rows] * dX) < 4e-3
    ref_dg = torch.zeros(Bs, d, dtype=torch.float64).index_add_(0, rows, (dX * delta).double())
    assert rel_linf(dmod.cpu()[:, 2 * d:3 * d], ref_dg) < 1e-5
    assert float(dmod.cpu()[:, :2 * d].abs().max()) == 0.0
    # --- LayerNorm + modulate backward
    xr = x1.double().requires_grad_(True)
    sh = mod[:, 3 * d:4 * d].double().requires_grad_(True)
    sc = mod[:, 4 * d:5 * d].double().requires_grad_(True)
    xn = do.layer_norm(xr) * (1 + sc[rows]) + sh[rows]
    dxn = bf16r(torch.randn((M, d), generator=gen))
    xn.backward(dxn.double())
    dXacc = dev(dX.clone())
    dmod2 = dev(torch.zeros_like(mod))
    N.check(N.lib().bsi_ln_mod_bwd(N.ptr(dev(dxn.to(torch.bfloat16))), N.ptr(dev(x1)), dmod_in.data_ptr() + 16 * d, 6 * d,
                                   dmod2.data_ptr() + 12 * d, dmod2.data_ptr() + 16 * d, 6 * d, N.ptr(dXacc), M, d, tokens,
                                   1e-5, N.stream()))
    assert rel_linf(dXacc, dX.double() + xr.grad) < 2e-5
    assert rel_linf(dmod2.cpu()[:, 3 * d:4 * d], sh.grad) < 1e-5
    assert rel_linf(dmod2.cpu()[:, 4 * d:5 * d], sc.grad) < 1e-5


@pytest.mark.parametrize("Bs,tokens,d", [(3, 64, 128), (2, 256, 1024), (1, 128, 256)])
def test_fused_ln_gate_backward(N, Bs, tokens, d):
    """bsi_ln_gate_bwd (LayerNorm-modulate backward with saved statistics + gated-residual backward in one pass) against
    fp64 autograd of  xn = LN(x) * (1 + scale) + shift,  x = x_below + gate * delta  (dit.py:50-55,93-102)."""
    gen = torch.Generator().manual_seed(7 * Bs + tokens + d)
    M = Bs * tokens
    rows = torch.arange(M) // tokens
    x_below = torch.randn((M, d), generator=gen) * 2 + 0.5
    delta = bf16r(torch.randn((M, d), generator=gen))
    mod = torch.randn((Bs, 6 * d), generator=gen) * 0.3
    dX_in = torch.randn((M, d), generator=gen)
    dxn = bf16r(torch.randn((M, d), generator=gen))
    xb = x_below.double().requires_grad_(True)
    dl = delta.double().requires_grad_(True)
    gate = mod[:, 2 * d:3 * d].double().requires_grad_(True)
    sh = mod[:, 3 * d:4 * d].double().requires_grad_(True)
    sc = mod[:, 4 * d:5 * d].double().requires_grad_(True)
    x = torch.addcmul(xb, gate[rows], dl)
    x.retain_grad()
    xn = do.layer_norm(x) * (1 + sc[rows]) + sh[rows]
    ((xn * dxn.double()).sum() + (x * dX_in.double()).sum()).backward()
    # statistics exactly as the forward kernel writes them
    x32 = x.detach().float()
    mean = x32.mean(dim=1)
    rstd = 1.0 / torch.sqrt(((x32 - mean[:, None]) ** 2).mean(dim=1) + 1e-5)
    stats = torch.stack((mean, rstd), dim=1).contiguous()
    dmod_in = dev(mod)
    for mode in ("both", "ln", "gate"):
        dXacc = dev(dX_in.clone())
        dmod = dev(torch.zeros_like(mod))
        dd = empty(M, d, dtype=torch.bfloat16)
        dd.zero_()
        ln, gt = mode != "gate", mode != "ln"
        N.check(N.lib().bsi_ln_gate_bwd(
            N.ptr(dev(dxn.to(torch.bfloat16))) if ln else None, N.ptr(dev(x32)) if ln else None, N.ptr(dev(stats)) if ln else None,
            dmod_in.data_ptr() + 16 * d if ln else None, 6 * d, dmod.data_ptr() + 12 * d if ln else None,
            dmod.data_ptr() + 16 * d if ln else None, 6 * d, N.ptr(dXacc),
            N.ptr(dev(delta.to(torch.bfloat16))) if gt else None, dmod_in.data_ptr() + 8 * d if gt else None, 6 * d,
            dmod.data_ptr() + 8 * d if gt else None, 6 * d, N.ptr(dd) if gt else None, M, d, tokens, N.stream()))
        want_dx = x.grad if ln else dX_in.double()            # dL/dx: the residual-path gradient plus the LayerNorm's
        assert rel_linf(dXacc, want_dx) < 2e-5, (mode, rel_linf(dXacc, want_dx))
        if ln:
            assert rel_linf(dmod.cpu()[:, 3 * d:4 * d], sh.grad) < 1e-5 and rel_linf(dmod.cpu()[:, 4 * d:5 * d], sc.grad) < 1e-5
        if gt:
            assert rel_linf(dd.cpu().float(), gate.detach()[rows] * want_dx) < 4e-3, mode
            ref_dg = torch.zeros(Bs, d, dtype=torch.float64).index_add_(0, rows, want_dx * delta.double())
            assert rel_linf(dmod.cpu()[:, 2 * d:3 * d], ref_dg) < 1e-5, mode
        else:
            assert float(dmod.cpu()[:, 2 * d:3 * d].abs().max()) == 0.0
    assert rel_linf(gate.grad, torch.zeros(Bs, d, dtype=torch.float64).index_add_(0, rows, x.grad * delta.double())) < 1e-12


def test_cast_batch_writes_every_shadow_in_one_launch(N):
    """bsi_cast_batch_bf16: row-major (one of them padded with zero columns), transposed and both-at-once shadows of five matrices with
    ragged shapes in one launch, bit-identical to bsi_cast_bf16 / bsi_cast_transpose_bf16 per matrix."""
    gen = torch.Generator().manual_seed(5)
    shapes = [(300, 84, 96, True, False), (64, 64, 64, True, True), (1024, 1024, 1024, True, True), (130, 257, 257, False, True),
              (7, 5, 8, True, True)]
    lib = N.lib()
    srcs, plain, trans, descs, tiles = [], [], [], [], 0
    for rows, cols, ld, want_p, want_t in shapes:
        w = dev(torch.randn((rows, cols), generator=gen))
        o = torch.full((rows, ld), float("nan"), dtype=torch.bfloat16, device=DEV) if want_p else None
        t = torch.full((cols, rows), float("nan"), dtype=torch.bfloat16, device=DEV) if want_t else None
        descs.append(N.CastDesc(w.data_ptr(), o.data_ptr() if want_p else None, t.data_ptr() if want_t else None, rows, cols, ld, rows, tiles, 0))
        tiles += lib.bsi_cast_batch_tiles(rows, cols, ld if want_p else cols)
        srcs.append(w); plain.append(o); trans.append(t)
    arr = (N.CastDesc * len(descs))(*descs)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    N.check(lib.bsi_cast_batch_bf16(N.ptr(table), len(descs), tiles, N.stream()))
    for (rows, cols, ld, want_p, want_t), w, o, t in zip(shapes, srcs, plain, trans):
        if want_p:
            ref = torch.empty((rows, ld), dtype=torch.bfloat16, device=DEV)
            N.check(lib.bsi_cast_bf16(N.ptr(w), rows, cols, N.ptr(ref), ld, N.stream()))
            assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), (rows, cols)
            assert float(o[:, cols:].float().abs().sum()) == 0.0
        if want_t:
            ref = torch.empty((cols, rows), dtype=torch.bfloat16, device=DEV)
            N.check(lib.bsi_cast_transpose_bf16(N.ptr(w), rows, cols, N.ptr(ref), rows, N.stream()))
            assert torch.equal(t.view(torch.int16), ref.view(torch.int16)), (rows, cols)


def test_cast_transpose_and_silu_bwd(N):
    gen = torch.Generator().manual_seed(9)
    w = torch.randn((300, 84), generator=gen)
    out = empty(84, 320, dtype=torch.bfloat16)
    N.check(N.lib().bsi_cast_transpose_bf16(N.ptr(dev(w)), 300, 84, N.ptr(out), 320, N.stream()))
    ref = torch.zeros(84, 320)
    ref[:, :300] = w.t()
    assert torch.equal(out.cpu(), ref.to(torch.bfloat16))
    ds, pre = torch.randn(1000, generator=gen), torch.randn(1000, generator=gen) * 3
    o = empty(1000, dtype=torch.bfloat16)
    N.check(N.lib().bsi_silu_bwd_bf16(N.ptr(dev(ds)), N.ptr(dev(pre)), 1000, N.ptr(o), N.stream()))
    p = pre.double().requires_grad_(True)
    do.silu(p).backward(ds.double())
    assert rel_linf(o.cpu().float(), p.grad) < 5e-3


# ----------------------------------------------------------------------------------------------
# VDM-UNet building blocks
# ----------------------------------------------------------------------------------------------
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,Cin,Cin2,Cout,taps,epi", [
    (2, 8, 64, 0, 64, 9, "bias"), (3, 32, 128, 0, 128, 9, "film"), (2, 32, 128, 0, 128, 9, "resid"),
    (2, 32, 128, 256, 128, 9, "resid_skip"), (2, 32, 32, 0, 128, 9, "resid0"), (2, 32, 128, 0, 384, 9, "bias"),
    (9, 8, 256, 0, 64, 1, "bias"), (3, 16, 384, 0, 128, 9, "bias"), (5, 16, 256, 128, 128, 9, "resid_skip"),
    (12, 32, 128, 0, 128, 9, "bias_persistent"), (12, 32, 64, 0, 128, 9, "resid_persistent"), (3, 8, 128, 0, 384, 1, "resid"), (3, 32, 256, 0, 384, 9, "bias_ring"), (2, 32, 128, 0, 128, 9, "resid_ring"), (3, 32, 128, 0, 128, 9, "bias_slab"), (2, 32, 256, 0, 384, 9, "bias"), (5, 16, 128, 0, 256, 9, "resid"), (48, 16, 64, 0, 128, 9, "bias_persistent_slab"),
    (48, 16, 128, 0, 128, 9, "bias_persistent_ring"), (5, 8, 128, 0, 128, 9, "film"), (3, 32, 128, 0, 128, 9, "film_ring"),
    (6, 16, 256, 0, 128, 9, "film"),
    # 3 x 3 + folded 1 x 1 skip with two skip chunks per chunk (the up blocks' conv2): the two-source slab kernel -- several tiles per
    # workgroup, image width 16, two chunks, a ragged last tile, and the ring kernel on the same shape
    (12, 32, 128, 256, 128, 9, "resid_skip_persistent"), (5, 16, 128, 256, 128, 9, "resid_skip"), (3, 32, 64, 128, 128, 9, "resid_skip"),
    (3, 16, 128, 256, 256, 9, "resid_skip"), (2, 32, 128, 256, 128, 9, "resid_skip_ring"), (7, 16, 128, 256, 128, 9, "bias_skip_persistent")])
def test_conv_implicit_gemm(N, B, H, Cin, Cin2, Cout, taps, epi):
    if epi.endswith("_slab"):  # the pixel-slab kernel on a shape the dispatcher gives to the ring kernel
        N.check(N.lib().bsi_conv_set_ablation(512))
        try:
            return test_conv_implicit_gemm(N, B, H, Cin, Cin2, Cout, taps, epi[:-len("_slab")])
        finally:
            N.check(N.lib().bsi_conv_set_ablation(0))
    if epi.endswith("_ring"):  # the ring kernel on shapes the pixel-slab kernel would take
        N.check(N.lib().bsi_conv_set_ablation(256))
        try:
            return test_conv_implicit_gemm(N, B, H, Cin, Cin2, Cout, taps, epi[:-len("_ring")])
        finally:
            N.check(N.lib().bsi_conv_set_ablation(0))
    if epi.endswith("_persistent"):  # 8 workgroups walk 24 tiles: tile-to-tile hand-over of the DMA ring and the epilogue stores
        N.check(N.lib().bsi_conv_set_grid_limit(8))
        try:
            return test_conv_implicit_gemm(N, B, H, Cin, Cin2, Cout, taps, epi[:-len("_persistent")])
        finally:
            N.check(N.lib().bsi_conv_set_grid_limit(0))
    gen = torch.Generator().manual_seed(B + H + Cin + Cout)
    ks = 3 if taps == 9 else 1
    x = bf16r(torch.randn((B, Cin, H, H), generator=gen))
    w = bf16r(torch.randn((Cout, Cin, ks, ks), generator=gen) / math.sqrt(Cin * taps))
    bias = torch.randn(Cout, generator=gen)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=ks // 2)
    K = taps * Cin + Cin2
    wp = empty(Cout, K, dtype=torch.bfloat16)
    N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w)), Cout, Cin, taps, Cin, K, 0, N.ptr(wp), N.stream()))
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    a = N.ConvArgs(x=dev(_nhwc(x).to(torch.bfloat16)).data_ptr(), w=wp.data_ptr(), bias=dev(bias).data_ptr(),
                   zeros=zeros.data_ptr(), B=B, H=H, W=H, Cin=Cin, Cin2=Cin2, Cout=Cout, taps=taps, ldo=Cout)
    if Cin2:
        x2 = bf16r(torch.randn((B, Cin2, H, H), generator=gen))
        w2 = bf16r(torch.randn((Cout, Cin2, 1, 1), generator=gen) / math.sqrt(Cin2))
        N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w2)), Cout, Cin2, 1, Cin2, K, taps * Cin, N.ptr(wp), N.stream()))
        a.x2 = dev(_nhwc(x2).to(torch.bfloat16)).data_ptr()
        ref = ref + torch.nn.functional.conv2d(x2.double(), w2.double())
    if epi in ("bias", "bias_skip"):
        out = empty(B * H * H, Cout, dtype=torch.bfloat16)
        a.out, a.epilogue = out.data_ptr(), N.CONV_BIAS_BF16
        N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
        assert rel_linf(out.cpu().float(), _nhwc(ref).reshape(-1, Cout)) < 5e-3
    elif epi == "film":
        film = torch.randn((B, 2 * Cout), generator=gen) * 0.5
        out = empty(B * H * H, Cout, dtype=torch.bfloat16)
        a.out, a.epilogue, a.film, a.film_rows, a.film_stride = out.data_ptr(), N.CONV_FILM_SILU_BF16, dev(film).data_ptr(), B, 2 * Cout
        N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
        y = ref * (film[:, :Cout, None, None].double() + 1) + film[:, Cout:, None, None].double()
        assert rel_linf(out.cpu().float(), _nhwc(do.silu(y)).reshape(-1, Cout)) < 5e-3
    else:
        res = torch.randn((B * H * H, Cout), generator=gen) if epi == "resid" else None
        out = empty(B * H * H, Cout)
        a.out, a.epilogue = out.data_ptr(), N.CONV_BIAS_RESID_F32
        if res is not None:
            a.resid = dev(res).data_ptr()
        N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
        want = _nhwc(ref).reshape(-1, Cout) + (res.double() if res is not None else 0)
        assert rel_linf(out, want) < 3e-5, rel_linf(out, want)


@pytest.mark.parametrize("epi_name", ["bias", "resid"])
def test_conv_full_size_slab_against_ring(N, epi_name):
    """256 images (512 tiles on 256 CUs, two per persistent workgroup) at the UNet's up-block shape 256 -> 128: the pixel-slab
    kernel against the ring kernel on the whole output, each run twice (reproducible), and against fp64 on one image."""
    B, H, Cin, Cout = 256, 32, 256, 128
    gen = torch.Generator().manual_seed(5)
    M, K = B * H * H, 9 * Cin
    x = torch.randn((M, Cin), generator=gen).to(torch.bfloat16)
    w = bf16r(torch.randn((Cout, Cin, 3, 3), generator=gen) / math.sqrt(K))
    bias = torch.randn(Cout, generator=gen)
    xd, bd = dev(x), dev(bias)
    wp = empty(Cout, K, dtype=torch.bfloat16)
    N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w)), Cout, Cin, 9, Cin, K, 0, N.ptr(wp), N.stream()))
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    f32 = epi_name == "resid"
    res = dev(torch.randn((M, Cout), generator=gen)) if f32 else None
    outs = {}
    try:
        for abl in (512, 256):  # 512 = slab kernel, 256 = ring kernel
            N.check(N.lib().bsi_conv_set_ablation(abl))
            for rep in range(2):
                out = torch.full((M, Cout), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
                a = N.ConvArgs(x=xd.data_ptr(), w=wp.data_ptr(), bias=bd.data_ptr(), zeros=zeros.data_ptr(), B=B, H=H, W=H, Cin=Cin, Cin2=0,
                               Cout=Cout, taps=9, ldo=Cout, out=out.data_ptr(),
                               epilogue=N.CONV_BIAS_RESID_F32 if f32 else N.CONV_BIAS_BF16)
                if f32:
                    a.resid = res.data_ptr()
                N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
                torch.cuda.synchronize()
                if rep:
                    assert torch.equal(out, outs[abl]), f"kernel choice {abl} is not reproducible"
                outs[abl] = out
    finally:
        N.check(N.lib().bsi_conv_set_ablation(0))
    a0, a1 = outs[512].float(), outs[256].float()
    assert bool(torch.isfinite(a0).all())
    # different fp32 summation orders (chunk-major vs tap-major K), bf16 rounding on top for the bf16 epilogue
    tol = 2e-5 if f32 else 1.6e-2
    assert float(((a0 - a1).abs() / (a1.abs() + 0.05)).max()) < tol * (1 if not f32 else 50)
    img = 171  # one image against fp64 (second tile of a workgroup)
    xi = x[img * H * H:(img + 1) * H * H].double().reshape(1, H, H, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xi, w.double(), bias.double(), padding=1)
    ref = _nhwc(ref).reshape(-1, Cout)
    if f32:
        ref = ref + res[img * H * H:(img + 1) * H * H].cpu().double()
    assert rel_linf(a0[img * H * H:(img + 1) * H * H].cpu(), ref) < (3e-5 if f32 else 5e-3)


def test_conv_full_size_two_source_slab_against_ring(N):
    """conv2 of the up blocks at full size (256 images of 32 x 32: 512 tiles on 256 CUs; 3 x 3 over 128 channels + the folded 1 x 1
    skip convolution over the 256-channel concatenation, fp32 output with GroupNorm partials): the two-source slab kernel against the
    ring kernel on the whole output and the partials, each run twice (reproducible), and against fp64 on one image."""
    B, H, Cin, Cin2, Cout = 256, 32, 128, 256, 128
    gen = torch.Generator().manual_seed(6)
    M, K = B * H * H, 9 * Cin + Cin2
    x = torch.randn((M, Cin), generator=gen).to(torch.bfloat16)
    x2 = torch.randn((M, Cin2), generator=gen).to(torch.bfloat16)
    w = bf16r(torch.randn((Cout, Cin, 3, 3), generator=gen) / math.sqrt(9 * Cin))
    w2 = bf16r(torch.randn((Cout, Cin2, 1, 1), generator=gen) / math.sqrt(Cin2))
    bias = torch.randn(Cout, generator=gen)
    xd, x2d, bd = dev(x), dev(x2), dev(bias)
    wp = empty(Cout, K, dtype=torch.bfloat16)
    N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w)), Cout, Cin, 9, Cin, K, 0, N.ptr(wp), N.stream()))
    N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w2)), Cout, Cin2, 1, Cin2, K, 9 * Cin, N.ptr(wp), N.stream()))
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    outs, parts = {}, {}
    try:
        for abl in (0, 256):  # 0 = the dispatcher's choice (two-source slab kernel), 256 = ring kernel
            N.check(N.lib().bsi_conv_set_ablation(abl))
            for rep in range(2):
                out = torch.full((M, Cout), float("nan"), dtype=torch.float32, device=DEV)
                gp = torch.full((M // 128, Cout // 4, 2), float("nan"), dtype=torch.float32, device=DEV)
                a = N.ConvArgs(x=xd.data_ptr(), x2=x2d.data_ptr(), w=wp.data_ptr(), bias=bd.data_ptr(), zeros=zeros.data_ptr(), B=B, H=H, W=H,
                               Cin=Cin, Cin2=Cin2, Cout=Cout, taps=9, ldo=Cout, out=out.data_ptr(), epilogue=N.CONV_BIAS_RESID_F32,
                               gn_partial=gp.data_ptr())
                N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
                torch.cuda.synchronize()
                if rep:
                    assert torch.equal(out, outs[abl]) and torch.equal(gp, parts[abl]), f"kernel choice {abl} is not reproducible"
                outs[abl], parts[abl] = out, gp
    finally:
        N.check(N.lib().bsi_conv_set_ablation(0))
    a0, a1 = outs[0], outs[256]
    assert bool(torch.isfinite(a0).all()) and bool(torch.isfinite(parts[0]).all())
    assert float(((a0 - a1).abs() / (a1.abs() + 0.05)).max()) < 1e-3      # different fp32 summation orders of the same products
    assert float(((parts[0][..., 0] - parts[256][..., 0]).abs()).max()) < 1e-4
    for img in (0, 171, 255):  # first tile, second tile of a workgroup, last tile: against fp64
        sl = slice(img * H * H, (img + 1) * H * H)
        xi = x[sl].double().reshape(1, H, H, Cin).permute(0, 3, 1, 2)
        x2i = x2[sl].double().reshape(1, H, H, Cin2).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi, w.double(), bias.double(), padding=1) + torch.nn.functional.conv2d(x2i, w2.double())
        assert rel_linf(a0[sl].cpu(), _nhwc(ref).reshape(-1, Cout)) < 3e-5


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(3, 64, 64, 0, 1), (2, 1024, 128, 0, 1), (2, 1024, 128, 128, 1), (2, 1024, 128, 0, 0)])
def test_groupnorm_nhwc(N, B, HW, C1, C2, silu):
    gen = torch.Generator().manual_seed(B + HW + C1 + C2)
    Cc = C1 + C2
    x1 = torch.randn((B, HW, C1), generator=gen) * 2 + 0.5
    x2 = torch.randn((B, HW, C2), generator=gen) if C2 else None
    ga, be = torch.randn(Cc, generator=gen), torch.randn(Cc, generator=gen)
    xc = torch.cat([x1, x2], 2) if C2 else x1
    side = int(math.isqrt(HW))
    xn = xc.permute(0, 2, 1).reshape(B, Cc, side, side).double()
    from oracle.unet_oracle import group_norm
    ref = group_norm(xn, 32, ga.double(), be.double())
    if silu:
        ref = do.silu(ref)
    ref = ref.reshape(B, Cc, HW).permute(0, 2, 1)
    out = empty(B, HW, Cc, dtype=torch.bfloat16)
    raw = empty(B, HW, Cc, dtype=torch.bfloat16)
    N.check(N.lib().bsi_groupnorm_nhwc(N.ptr(dev(x1)), C1, N.ptr(dev(x2)) if C2 else None, C2, B, HW, N.ptr(dev(ga)),
                                       N.ptr(dev(be)), 1e-5, silu, N.ptr(out), N.ptr(raw), N.stream()))
    assert float((out.cpu().double() - ref).abs().max()) <= float(ref.abs().max()) * 2 ** -8 + 1e-3
    assert torch.equal(raw.cpu(), xc.to(torch.bfloat16))


@pytest.mark.parametrize("B,H,Cin,Cin2,cat,kern,offset", [
    (3, 32, 128, 0, False, "slab", 0.0), (3, 32, 128, 0, True, "ring", 0.0), (2, 32, 128, 256, True, "ring", 0.0),
    (5, 16, 128, 0, True, "slab", 0.0), (2, 32, 128, 0, False, "slab", 300.0), (12, 32, 128, 0, True, "persistent", 0.0)])
def test_conv_groupnorm_partials_and_apply(N, B, H, Cin, Cin2, cat, kern, offset):
    """GroupNorm statistics from the convolution epilogue (bsi_conv_args.gn_partial) + the streaming normalisation
    (bsi_groupnorm_apply_nhwc) against fp64 GroupNorm of the convolution's own fp32 output (residual_block.py:42-43): partial
    (mean, M2) per 128 pixels x 4 channels, merged statistics, normalised bf16 output, raw bf16 copy; `offset` shifts the
    residual so that |mean| >> std (the pivot keeps M2 exact); `cat` normalises cat(x1, x2) with 8-channel groups."""
    from oracle.unet_oracle import group_norm
    Cout, HW, M = 128, H * H, B * H * H
    gen = torch.Generator().manual_seed(B + H + Cin + Cin2 + int(offset))
    K = 9 * Cin + Cin2
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    N.check(N.lib().bsi_conv_set_ablation(256 if kern == "ring" else 512))
    if kern == "persistent":
        N.check(N.lib().bsi_conv_set_grid_limit(8))
    try:
        maps, parts = [], []
        for which in range(2 if cat else 1):
            x = torch.randn((M, Cin), generator=gen).to(torch.bfloat16)
            w = bf16r(torch.randn((Cout, Cin, 3, 3), generator=gen) / math.sqrt(9 * Cin))
            bias = torch.randn(Cout, generator=gen)
            res = torch.randn((M, Cout), generator=gen) * (1 + which) + offset
            wp = empty(Cout, K, dtype=torch.bfloat16)
            N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w)), Cout, Cin, 9, Cin, K, 0, N.ptr(wp), N.stream()))
            out = torch.full((M, Cout), float("nan"), device=DEV)
            part = torch.full((M // 128, Cout // 4, 2), float("nan"), device=DEV)
            xd, bd, rd = dev(x), dev(bias), dev(res)
            a = N.ConvArgs(x=xd.data_ptr(), w=wp.data_ptr(), bias=bd.data_ptr(), zeros=zeros.data_ptr(), B=B, H=H, W=H, Cin=Cin, Cin2=Cin2,
                           Cout=Cout, taps=9, ldo=Cout, out=out.data_ptr(), epilogue=N.CONV_BIAS_RESID_F32, resid=rd.data_ptr(),
                           gn_partial=part.data_ptr())
            if Cin2:
                x2 = torch.randn((M, Cin2), generator=gen).to(torch.bfloat16)
                w2 = bf16r(torch.randn((Cout, Cin2, 1, 1), generator=gen) / math.sqrt(Cin2))
                N.check(N.lib().bsi_conv_weight_pack(N.ptr(dev(w2)), Cout, Cin2, 1, Cin2, K, 9 * Cin, N.ptr(wp), N.stream()))
                x2d = dev(x2)
                a.x2 = x2d.data_ptr()
            N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
            # the same convolution without the statistics: identical output
            out0 = torch.empty_like(out)
            a.out, a.gn_partial = out0.data_ptr(), None
            N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
            torch.cuda.synchronize()
            assert torch.equal(out, out0)
            if which == 0:  # without a residual (the encode convolution's epilogue): the same statistics path, no row fetches
                out1, part1 = torch.full_like(out, float("nan")), torch.full_like(part, float("nan"))
                a.out, a.gn_partial, a.resid = out1.data_ptr(), part1.data_ptr(), None
                N.check(N.lib().bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
                torch.cuda.synchronize()
                assert float((out1.double() - (out.double() - rd.double())).abs().max()) < 1e-4 * (1 + abs(offset))
                p64 = out1.cpu().double().reshape(M // 128, 128, Cout // 4, 4)
                assert float((part1.cpu().double()[..., 0] - p64.mean(dim=(1, 3))).abs().max()) < 1e-5
            o64 = out.cpu().double().reshape(M // 128, 128, Cout // 4, 4)
            mean = o64.mean(dim=(1, 3))
            m2 = ((o64 - mean[:, None, :, None]) ** 2).sum(dim=(1, 3))
            pc = part.cpu().double()
            assert bool(torch.isfinite(pc).all())
            assert float((pc[..., 0] - mean).abs().max()) < 1e-5 * (1 + abs(offset))
            assert float(((pc[..., 1] - m2).abs() / m2).max()) < 2e-4, float(((pc[..., 1] - m2).abs() / m2).max())
            maps.append(out)
            parts.append(part)
    finally:
        N.check(N.lib().bsi_conv_set_ablation(0))
        N.check(N.lib().bsi_conv_set_grid_limit(0))
    Cc = Cout * len(maps)
    ga, be = torch.randn(Cc, generator=gen), torch.randn(Cc, generator=gen)
    xc = torch.cat([m.cpu() for m in maps], 1).reshape(B, HW, Cc)
    xn = xc.permute(0, 2, 1).reshape(B, Cc, H, H).double()
    for silu in (1, 0):
        ref = group_norm(xn, 32, ga.double(), be.double())
        if silu:
            ref = do.silu(ref)
        ref = ref.reshape(B, Cc, HW).permute(0, 2, 1)
        outn = torch.full((B, HW, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
        raw = torch.full((B, HW, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
        stats = torch.full((B, 32, 2), float("nan"), device=DEV)
        gad, bed = dev(ga), dev(be)
        N.check(N.lib().bsi_groupnorm_apply_nhwc(N.ptr(maps[0]), Cout, N.ptr(parts[0]), N.ptr(maps[1]) if cat else None, Cout if cat else 0,
                                                 N.ptr(parts[1]) if cat else None, B, HW, N.ptr(gad), N.ptr(bed), 1e-5, silu, N.ptr(outn),
                                                 N.ptr(raw), N.ptr(stats), N.stream()))
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        assert float((outn.cpu().double() - ref).abs().max()) <= scale * 2 ** -8 + 1e-3
        assert torch.equal(raw.cpu(), xc.to(torch.bfloat16))
        g = xn.reshape(B, 32, -1)
        st = stats.cpu().double()
        assert float((st[..., 0] - g.mean(2)).abs().max()) < 1e-5 * (1 + abs(offset))
        rstd = 1.0 / torch.sqrt(g.var(2, unbiased=False) + 1e-5)
        assert float(((st[..., 1] - rstd).abs() / rstd).max()) < 1e-4
        # against the reduce-then-normalise kernel on the same input: at most one bf16 rounding apart
        old = empty(B, HW, Cc, dtype=torch.bfloat16)
        N.check(N.lib().bsi_groupnorm_nhwc(N.ptr(maps[0]), Cout, N.ptr(maps[1]) if cat else None, Cout if cat else 0, B, HW, N.ptr(gad),
                                           N.ptr(bed), 1e-5, silu, N.ptr(old), None, N.stream()))
        assert float((old.float() - outn.float()).abs().max()) <= scale * 2 ** -7 + 1e-3


def test_conv_weight_pack_batch(N):
    """bsi_conv_weight_pack_batch (all convolution weights of a model in one launch) against the per-convolution entry points:
    3x3 with channel padding (the encoder: 21 -> 32), a 3x3 with the 1x1 skip weights appended behind it, a 1x1, and the
    transposed (input-gradient) layout; bit-equal."""
    gen = torch.Generator().manual_seed(7)
    convs = [(128, 21, 9, 32, None), (128, 128, 9, 128, 256), (64, 256, 1, 256, None), (384, 128, 9, 128, None)]
    keep, descs, want = [], [], []
    for cout, cin, taps, cin_pad, extra in convs:
        ks = 3 if taps == 9 else 1
        w = dev(torch.randn((cout, cin, ks, ks), generator=gen))
        k = taps * cin_pad + (extra or 0)
        out = torch.full((cout, k), float("nan"), dtype=torch.bfloat16, device=DEV)
        ref = torch.full((cout, k), float("nan"), dtype=torch.bfloat16, device=DEV)
        descs.append(N.ConvPackDesc(w.data_ptr(), out.data_ptr(), cout, cin, taps, cin_pad, k, 0))
        N.check(N.lib().bsi_conv_weight_pack(N.ptr(w), cout, cin, taps, cin_pad, k, 0, N.ptr(ref), N.stream()))
        keep.append(w)
        if extra:
            w2 = dev(torch.randn((cout, extra, 1, 1), generator=gen))
            descs.append(N.ConvPackDesc(w2.data_ptr(), out.data_ptr(), cout, extra, 1, extra, k, taps * cin_pad))
            N.check(N.lib().bsi_conv_weight_pack(N.ptr(w2), cout, extra, 1, extra, k, taps * cin_pad, N.ptr(ref), N.stream()))
            keep.append(w2)
        want.append((out, ref))
    darr = (N.ConvPackDesc * len(descs))(*descs)
    ddev = torch.frombuffer(bytearray(bytes(darr)), dtype=torch.uint8).to(DEV)
    N.check(N.lib().bsi_conv_weight_pack_batch(N.ptr(ddev), len(descs), 0, N.stream()))
    torch.cuda.synchronize()
    for out, ref in want:
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    # transposed layout
    tdescs, twant = [], []
    for w in keep[:2] + keep[3:]:
        cout, cin, kh, kw = w.shape
        taps = kh * kw
        out = torch.full((cin, taps * cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        ref = torch.empty_like(out)
        tdescs.append(N.ConvPackDesc(w.data_ptr(), out.data_ptr(), cout, cin, taps, cin, taps * cout, 0))
        N.check(N.lib().bsi_conv_weight_pack_t(N.ptr(w), cout, cin, taps, taps * cout, N.ptr(ref), N.stream()))
        twant.append((out, ref))
    darr = (N.ConvPackDesc * len(tdescs))(*tdescs)
    ddev = torch.frombuffer(bytearray(bytes(darr)), dtype=torch.uint8).to(DEV)
    N.check(N.lib().bsi_conv_weight_pack_batch(N.ptr(ddev), len(tdescs), 1, N.stream()))
    torch.cuda.synchronize()
    for out, ref in twant:
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("B,H,W,Cin,Cin2,Cout,taps", [(2, 8, 8, 128, 0, 128, 9), (3, 16, 16, 256, 256, 128, 9),
                                                     (2, 16, 8, 32, 0, 128, 9), (2, 8, 8, 128, 0, 384, 9),
                                                     (5, 16, 16, 384, 0, 128, 9), (2, 8, 8, 256, 0, 128, 1),
                                                     (3, 32, 32, 128, 0, 128, 9), (2, 32, 32, 256, 0, 128, 9), (2, 32, 32, 128, 0, 384, 9),
                                                     (7, 8, 32, 128, 0, 128, 9), (2, 32, 32, 128, 256, 128, 9)])
def test_conv_weight_and_input_gradients(N, B, H, W, Cin, Cin2, Cout, taps):
    """bsi_conv_wgrad_nhwc_bf16 (+ unpack) and the input-gradient convolution (bsi_conv_weight_pack_t + the forward
    kernel) against autograd of torch.nn.functional.conv2d in float64."""
    gen = torch.Generator().manual_seed(B + H + Cin + Cout + taps)
    ks = 3 if taps == 9 else 1
    x = bf16r(torch.randn((B, Cin, H, W), generator=gen)).double().requires_grad_(True)
    w = bf16r(torch.randn((Cout, Cin, ks, ks), generator=gen) / math.sqrt(Cin * taps)).double().requires_grad_(True)
    dy = bf16r(torch.randn((B, Cout, H, W), generator=gen))
    y = torch.nn.functional.conv2d(x, w, padding=ks // 2)
    x2 = w2 = None
    if Cin2:
        x2 = bf16r(torch.randn((B, Cin2, H, W), generator=gen)).double()
        w2 = bf16r(torch.randn((Cout, Cin2, 1, 1), generator=gen) / math.sqrt(Cin2)).double().requires_grad_(True)
        y = y + torch.nn.functional.conv2d(x2, w2)
    y.backward(dy.double())
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    M, K = B * H * W, taps * Cin + Cin2
    dyd = dev(_nhwc(dy).reshape(M, Cout).to(torch.bfloat16))
    xd = dev(_nhwc(x.detach().float()).reshape(M, Cin).to(torch.bfloat16))
    x2d = dev(_nhwc(x2.float()).reshape(M, Cin2).to(torch.bfloat16)) if Cin2 else None
    lib = N.lib()
    ws = empty(lib.bsi_conv_wgrad_workspace_bytes(M, Cin, Cin2, Cout, taps), dtype=torch.uint8)
    packed = empty(Cout, K)
    N.check(lib.bsi_conv_wgrad_nhwc_bf16(N.ptr(dyd), Cout, N.ptr(xd), N.ptr(x2d) if Cin2 else None, N.ptr(zeros), B, H, W, Cin,
                                         Cin2, Cout, taps, N.ptr(packed), 0, N.ptr(ws), N.stream()))
    gw = empty(Cout, Cin, ks, ks)
    N.check(lib.bsi_conv_wgrad_unpack(N.ptr(packed), Cout, Cin, taps, Cin, K, 0, 0, N.ptr(gw), N.stream()))
    assert rel_linf(gw, w.grad) < 2e-5, rel_linf(gw, w.grad)
    if Cin2:
        gw2 = empty(Cout, Cin2, 1, 1)
        N.check(lib.bsi_conv_wgrad_unpack(N.ptr(packed), Cout, Cin2, 1, Cin2, K, taps * Cin, 0, N.ptr(gw2), N.stream()))
        assert rel_linf(gw2, w2.grad) < 2e-5
    # fused bias gradient: dbias[co] = sum over pixels of dY (autograd of the conv bias), same weight gradient
    if Cout % 4 == 0:
        packed_b, dbias = empty(Cout, K), empty(Cout)
        N.check(lib.bsi_conv_wgrad_bias_nhwc_bf16(N.ptr(dyd), Cout, N.ptr(xd), N.ptr(x2d) if Cin2 else None, N.ptr(zeros), B, H, W, Cin,
                                                  Cin2, Cout, taps, N.ptr(packed_b), N.ptr(dbias), 0, N.ptr(ws), N.stream()))
        assert torch.equal(packed_b.cpu(), packed.cpu())
        want_b = dyd.cpu().double().sum(0)
        assert rel_linf(dbias, want_b) < 2e-5, rel_linf(dbias, want_b)
    # accumulate flag
    N.check(lib.bsi_conv_wgrad_nhwc_bf16(N.ptr(dyd), Cout, N.ptr(xd), N.ptr(x2d) if Cin2 else None, N.ptr(zeros), B, H, W, Cin,
                                         Cin2, Cout, taps, N.ptr(packed), 1, N.ptr(ws), N.stream()))
    N.check(lib.bsi_conv_wgrad_unpack(N.ptr(packed), Cout, Cin, taps, Cin, K, 0, 1, N.ptr(gw), N.stream()))
    assert rel_linf(gw, 3 * w.grad) < 2e-5
    # input gradient = convolution of dY with the rotated, channel-swapped weights
    if Cout % 32 == 0 and Cin % 16 == 0:
        wt = empty(Cin, taps * Cout, dtype=torch.bfloat16)
        N.check(lib.bsi_conv_weight_pack_t(N.ptr(dev(w.detach().float())), Cout, Cin, taps, taps * Cout, N.ptr(wt), N.stream()))
        dx = empty(M, Cin)
        a = N.ConvArgs(x=dyd.data_ptr(), w=wt.data_ptr(), zeros=zeros.data_ptr(), out=dx.data_ptr(), B=B, H=H, W=W, Cin=Cout,
                       Cin2=0, Cout=Cin, taps=taps, ldo=Cin, epilogue=N.CONV_BIAS_RESID_F32)
        N.check(lib.bsi_conv_nhwc_bf16(C.byref(a), N.stream()))
        assert rel_linf(dx, _nhwc(x.grad).reshape(M, Cin)) < 2e-5, rel_linf(dx, _nhwc(x.grad).reshape(M, Cin))


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(3, 64, 64, 0, 1), (2, 1024, 128, 0, 1), (2, 256, 128, 128, 1), (2, 1024, 128, 0, 0)])
def test_groupnorm_backward(N, B, HW, C1, C2, silu):
    from oracle.unet_oracle import group_norm
    gen = torch.Generator().manual_seed(B + HW + C1 + C2 + 7)
    Cc = C1 + C2
    side = int(math.isqrt(HW))
    x1 = torch.randn((B, HW, C1), generator=gen) * 2 + 0.5
    x2 = torch.randn((B, HW, C2), generator=gen) if C2 else None
    ga, be = torch.randn(Cc, generator=gen), torch.randn(Cc, generator=gen)
    da = bf16r(torch.randn((B, HW, Cc), generator=gen))
    add = torch.randn((B, HW, Cc), generator=gen)
    add_b = torch.randn((B, HW, C1), generator=gen)
    xc = (torch.cat([x1, x2], 2) if C2 else x1).double().requires_grad_(True)
    gad, bed = ga.double().requires_grad_(True), be.double().requires_grad_(True)
    y = group_norm(xc.permute(0, 2, 1).reshape(B, Cc, side, side), 32, gad, bed)
    if silu:
        y = do.silu(y)
    y.backward(da.double().permute(0, 2, 1).reshape(B, Cc, side, side))
    want = xc.grad + add.double()
    want[:, :, :C1] += add_b.double()
    out1, out2 = empty(B, HW, C1), (empty(B, HW, C2) if C2 else None)
    dg, db = torch.ones(Cc, device=DEV), torch.ones(Cc, device=DEV)  # accumulated on top of existing values
    N.check(N.lib().bsi_groupnorm_bwd_nhwc(N.ptr(dev(da.to(torch.bfloat16))), N.ptr(dev(x1)), C1, N.ptr(dev(x2)) if C2 else None, C2,
                                           B, HW, N.ptr(dev(ga)), N.ptr(dev(be)), 1e-5, silu, N.ptr(dev(add)), N.ptr(dev(add_b)),
                                           N.ptr(out1), N.ptr(out2) if C2 else None, N.ptr(dg), N.ptr(db), N.stream()))
    assert rel_linf(out1, want[:, :, :C1]) < 2e-5, rel_linf(out1, want[:, :, :C1])
    if C2:
        assert rel_linf(out2, want[:, :, C1:]) < 2e-5
    assert rel_linf(dg - 1, gad.grad) < 2e-5 and rel_linf(db - 1, bed.grad) < 2e-5


@pytest.mark.parametrize("B,C1,C2,silu", [(3, 128, 0, 1), (2, 128, 128, 1), (2, 128, 128, 0)])
def test_groupnorm_backward_with_saved_statistics(N, B, C1, C2, silu):
    """bsi_groupnorm_bwd_cast_nhwc on 32 x 32 images with the forward's (mean, rstd) -- the form the training engine calls -- against
    fp64 autograd AND against bsi_groupnorm_bwd_nhwc (statistics recomputed) on the same inputs; the bf16 copy of the x1 gradient is
    the rounded fp32 one."""
    from oracle.unet_oracle import group_norm
    HW, side = 1024, 32
    gen = torch.Generator().manual_seed(B + C1 + C2 + silu)
    Cc = C1 + C2
    x1 = torch.randn((B, HW, C1), generator=gen) * 2 + 0.5
    x2 = torch.randn((B, HW, C2), generator=gen) if C2 else None
    ga, be = torch.randn(Cc, generator=gen), torch.randn(Cc, generator=gen)
    da = bf16r(torch.randn((B, HW, Cc), generator=gen))
    add = torch.randn((B, HW, Cc), generator=gen)
    add_b = torch.randn((B, HW, C1), generator=gen)
    xcat = torch.cat([x1, x2], 2) if C2 else x1
    xc = xcat.double().requires_grad_(True)
    gad, bed = ga.double().requires_grad_(True), be.double().requires_grad_(True)
    y = group_norm(xc.permute(0, 2, 1).reshape(B, Cc, side, side), 32, gad, bed)
    if silu:
        y = do.silu(y)
    y.backward(da.double().permute(0, 2, 1).reshape(B, Cc, side, side))
    want = xc.grad + add.double()
    want[:, :, :C1] += add_b.double()
    xg = xcat.reshape(B, HW, 32, Cc // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)   # [image, group, elements], fp32 as the forward sees them
    mean = xg.mean(dim=2)
    rstd = 1.0 / torch.sqrt(((xg - mean[:, :, None]) ** 2).mean(dim=2) + 1e-5)
    stats = dev(torch.stack((mean, rstd), dim=2).contiguous())
    dda, dx1, dx2 = dev(da.to(torch.bfloat16)), dev(x1), (dev(x2) if C2 else None)
    dga, dbe, dadd, daddb = dev(ga), dev(be), dev(add), dev(add_b)

    def run(fn, extra):
        out1, out2 = empty(B, HW, C1), (empty(B, HW, C2) if C2 else None)
        dg, db = torch.ones(Cc, device=DEV), torch.ones(Cc, device=DEV)  # accumulated on top of existing values
        N.check(fn(N.ptr(dda), N.ptr(dx1), C1, N.ptr(dx2) if C2 else None, C2, B, HW, N.ptr(dga), N.ptr(dbe), 1e-5, silu, N.ptr(dadd),
                   N.ptr(daddb), N.ptr(out1), N.ptr(out2) if C2 else None, N.ptr(dg), N.ptr(db), *extra, N.stream()))
        return out1, out2, dg, db
    obf = empty(B, HW, C1, dtype=torch.bfloat16)
    o1, o2, dg, db = run(N.lib().bsi_groupnorm_bwd_cast_nhwc, (N.ptr(obf), N.ptr(stats)))
    r1, r2, rg, rb = run(N.lib().bsi_groupnorm_bwd_nhwc, ())
    assert rel_linf(o1, want[:, :, :C1]) < 2e-5, rel_linf(o1, want[:, :, :C1])
    assert rel_linf(o1, r1.cpu().double()) < 5e-6
    if C2:
        assert rel_linf(o2, want[:, :, C1:]) < 2e-5 and rel_linf(o2, r2.cpu().double()) < 5e-6
    assert rel_linf(dg - 1, gad.grad) < 2e-5 and rel_linf(db - 1, bed.grad) < 2e-5
    assert torch.equal(obf.view(torch.int16), o1.to(torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("B,HW,Nc,p", [(3, 64, 64, 0.0), (2, 1024, 128, 0.1)])
def test_film_silu_dropout_forward_backward(N, B, HW, Nc, p):
    gen = torch.Generator().manual_seed(B + HW + Nc)
    M = B * HW
    h1 = bf16r(torch.randn((M, Nc), generator=gen))
    film = torch.randn((B, 2 * Nc), generator=gen) * 0.5
    dy = bf16r(torch.randn((M, Nc), generator=gen))
    seed, site = 1234, 5
    lib = N.lib()
    keep = torch.ones(M * Nc, dtype=torch.uint8, device=DEV)
    if p > 0:
        N.check(lib.bsi_dropout_mask(p, seed, site, M, Nc, N.ptr(keep), N.stream()))
        frac = float(keep.float().mean())
        assert abs(frac - (1 - p)) < 0.01, frac
    mask = keep.cpu().double().reshape(M, Nc) / (1 - p)
    hd = h1.double().requires_grad_(True)
    fd = film.double().requires_grad_(True)
    sc = fd[:, :Nc].repeat_interleave(HW, 0)
    sh = fd[:, Nc:].repeat_interleave(HW, 0)
    y = do.silu(hd * (sc + 1) + sh) * mask
    y.backward(dy.double())
    yb = empty(M, Nc, dtype=torch.bfloat16)
    h1d, fdv = dev(h1.to(torch.bfloat16)), dev(film)
    N.check(lib.bsi_film_silu(N.ptr(h1d), M, Nc, HW, N.ptr(fdv), B, 2 * Nc, p, seed, site, N.ptr(yb), N.stream()))
    assert rel_linf(yb.float(), y.detach()) < 5e-3
    dh1 = empty(M, Nc, dtype=torch.bfloat16)
    dfilm = torch.zeros((B, 2 * Nc), device=DEV)
    N.check(lib.bsi_film_silu_bwd(N.ptr(dev(dy.to(torch.bfloat16))), N.ptr(h1d), M, Nc, HW, N.ptr(fdv), B, 2 * Nc, p, seed, site,
                                  N.ptr(dh1), N.ptr(dfilm), 2 * Nc, N.stream()))
    assert rel_linf(dh1.float(), hd.grad) < 5e-3
    assert rel_linf(dfilm, fd.grad) < 1e-4, rel_linf(dfilm, fd.grad)


def test_unet_decode_backward(N):
    gen = torch.Generator().manual_seed(11)
    B, HW, Cc, Co = 3, 320, 128, 3
    h = torch.randn((B * HW, Cc), generator=gen)
    w = torch.randn((Co, Cc), generator=gen) / math.sqrt(Cc)
    g = torch.randn((B, Co, HW), generator=gen)
    c_out = torch.rand(B, generator=gen) + 0.5
    hd, wd = h.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = torch.zeros(Co, dtype=torch.double, requires_grad=True)
    f = (hd @ wd.t() + bd).reshape(B, HW, Co).permute(0, 2, 1)
    (f * c_out.double()[:, None, None]).backward(g.double())
    dh = empty(B * HW, Cc)
    dw, db = torch.zeros((Co, Cc), device=DEV), torch.zeros(Co, device=DEV)
    N.check(N.lib().bsi_unet_decode_bwd(N.ptr(dev(g)), N.ptr(dev(c_out)), 1, N.ptr(dev(h)), B, HW, Cc, N.ptr(dev(w)), Co, N.ptr(dh),
                                        N.ptr(dw), N.ptr(db), N.stream()))
    assert rel_linf(dh, hd.grad) < 1e-5 and rel_linf(dw, wd.grad) < 1e-5 and rel_linf(db, bd.grad) < 1e-5
