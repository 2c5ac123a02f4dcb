"""GPU parity of the individual C-ABI kernels against the oracle (CPU restatement of the PyG ops).

Tolerances: indices bit-exact; fp32 activations within atol 1e-5 + rtol 1e-5 of the oracle evaluated in
float64 (north_star: "bit-exact on indexing and within 1e-5 fp32 on activations/logits").
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import _lib  # noqa: E402
from oracle import pyg_ref  # noqa: E402

ATOL, RTOL = 1e-5, 1e-5


def dev():
    return torch.device("cuda:0")


def sp():
    return _lib.stream_ptr()


def build_plan(ei: torch.Tensor, n_src: int, n_dst: int):
    lib = _lib.require_device()
    E = ei.size(1)
    d = ei.device
    t = {
        "rowptr": torch.full((n_dst + 1,), -7, dtype=torch.int32, device=d),
        "col": torch.full((max(E, 1),), -7, dtype=torch.int32, device=d),
        "eid": torch.full((max(E, 1),), -7, dtype=torch.int32, device=d),
        "t_rowptr": torch.full((n_src + 1,), -7, dtype=torch.int32, device=d),
        "t_col": torch.full((max(E, 1),), -7, dtype=torch.int32, device=d),
        "t_pos": torch.full((max(E, 1),), -7, dtype=torch.int32, device=d),
    }
    plan = _lib.Plan(n_src, n_dst, E, t["rowptr"].data_ptr(), t["col"].data_ptr(), t["eid"].data_ptr(),
                     t["t_rowptr"].data_ptr(), t["t_col"].data_ptr(), t["t_pos"].data_ptr())
    scratch = torch.empty(max(int(lib.hmp_plan_scratch_bytes(E, n_src, n_dst)), 16), dtype=torch.uint8, device=d)
    status = torch.zeros(1, dtype=torch.int32, device=d)
    _lib.check(lib.hmp_plan_build(ei.data_ptr() if E > 0 else None, plan, scratch.data_ptr(), status.data_ptr(), sp()))
    torch.cuda.synchronize()
    t["status"] = int(status.item())
    t["plan"] = plan
    return t


def rand_edges(rng, E, n_src, n_dst):
    return torch.from_numpy(np.stack([rng.integers(0, max(n_src, 1), E), rng.integers(0, max(n_dst, 1), E)]).astype(np.int64))


@pytest.mark.parametrize("E,n_src,n_dst", [(0, 3, 4), (1, 1, 1), (17, 5, 3), (482, 62, 62), (20000, 700, 90), (5000, 1, 4000), (3000, 2000, 1),
                                          (60000, 2500, 3100), (100000, 3000, 2500), (200000, 9000, 12345), (70000, 8192, 4096)])
def test_plan_bit_exact(E, n_src, n_dst):
    rng = np.random.default_rng(E + n_src)
    ei = rand_edges(rng, E, n_src, n_dst)
    p = build_plan(ei.to(dev()), n_src, n_dst)
    assert p["status"] == 0
    src, dst = ei[0], ei[1]
    # oracle: stable sort by destination / by source
    order = torch.sort(dst, stable=True).indices
    rowptr = torch.zeros(n_dst + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_dst), 0)
    t_order = torch.sort(src, stable=True).indices
    t_rowptr = torch.zeros(n_src + 1, dtype=torch.int64)
    t_rowptr[1:] = torch.cumsum(torch.bincount(src, minlength=n_src), 0)
    assert torch.equal(p["rowptr"].cpu().long(), rowptr)
    assert torch.equal(p["t_rowptr"].cpu().long(), t_rowptr)
    if E:
        assert torch.equal(p["eid"].cpu().long()[:E], order)
        assert torch.equal(p["col"].cpu().long()[:E], src[order])
        assert torch.equal(p["t_col"].cpu().long()[:E], dst[t_order])
        pos_of = torch.empty(E, dtype=torch.int64)
        pos_of[order] = torch.arange(E)
        assert torch.equal(p["t_pos"].cpu().long()[:E], pos_of[t_order])


@pytest.mark.parametrize("group_by", ["dst", "src"])
@pytest.mark.parametrize("bad", [False, True])
def test_plan_bit_exact_on_row_grouped_edge_lists(group_by, bad, monkeypatch):
    """Edge lists written row by row (runs of 1..70 equal keys in neighbouring lanes, crossing wavefront boundaries, duplicate
    edges) take the wave-aggregated counters of ``plan_hist_kernel``: same stable CSR / CSC as a sort; out-of-range edges inside
    a run (holes) are dropped and flagged without disturbing their neighbours."""
    monkeypatch.setenv("HMP_PLAN_SMALL", "0")  # the multi-launch build with global counters
    rng = np.random.default_rng(11 if group_by == "dst" else 12)
    n_src, n_dst = 5000, 4000
    n_key = n_dst if group_by == "dst" else n_src
    keys = np.repeat(rng.permutation(n_key)[:2500], rng.integers(1, 71, 2500))  # grouped, not sorted: a key never comes back
    E = keys.size
    other = rng.integers(0, n_src if group_by == "dst" else n_dst, E)
    other[rng.random(E) < 0.2] = other[0]  # duplicates and long runs on the other side too
    src, dst = (other, keys) if group_by == "dst" else (keys, other)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    valid = torch.ones(E, dtype=torch.bool)
    if bad:
        holes = torch.from_numpy(rng.choice(E, 300, replace=False))
        ei[0, holes[:100]] = n_src  # one past the end
        ei[1, holes[100:200]] = -1
        ei[0, holes[200:]] = -5
        valid[holes] = False
    p = build_plan(ei.to(dev()), n_src, n_dst)
    assert (p["status"] & 1) == int(bad)
    ids = torch.nonzero(valid).flatten()
    s, d = ei[0, ids], ei[1, ids]
    order = ids[torch.sort(d, stable=True).indices]
    t_order = ids[torch.sort(s, stable=True).indices]
    rowptr = torch.zeros(n_dst + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(d, minlength=n_dst), 0)
    t_rowptr = torch.zeros(n_src + 1, dtype=torch.int64)
    t_rowptr[1:] = torch.cumsum(torch.bincount(s, minlength=n_src), 0)
    Ev = ids.numel()
    assert torch.equal(p["rowptr"].cpu().long(), rowptr)
    assert torch.equal(p["t_rowptr"].cpu().long(), t_rowptr)
    assert torch.equal(p["eid"].cpu().long()[:Ev], order)
    assert torch.equal(p["col"].cpu().long()[:Ev], ei[0][order])
    assert torch.equal(p["t_col"].cpu().long()[:Ev], ei[1][t_order])


def test_plan_multi_launch_and_single_launch_builds_agree(monkeypatch):
    """HMP_PLAN_SMALL pins the build: 0 = histogram/scan/fill/rank launches with global counters, 1 = one launch (LDS)"""
    rng = np.random.default_rng(77)
    for E, n_src, n_dst in [(482, 62, 62), (20000, 700, 90), (30000, 2300, 2300)]:
        ei = rand_edges(rng, E, n_src, n_dst).to(dev())
        monkeypatch.setenv("HMP_PLAN_SMALL", "0")
        a = build_plan(ei, n_src, n_dst)
        monkeypatch.setenv("HMP_PLAN_SMALL", "1")
        b = build_plan(ei, n_src, n_dst)
        monkeypatch.delenv("HMP_PLAN_SMALL")
        for k in ("rowptr", "col", "eid", "t_rowptr", "t_col", "t_pos"):
            assert torch.equal(a[k], b[k]), k


def test_plan_flags_out_of_range_edges():
    ei = torch.tensor([[0, 1, 9, 1], [0, 1, 1, -1]], dtype=torch.int64)
    p = build_plan(ei.to(dev()), 3, 2)
    assert p["status"] & 1
    assert p["rowptr"].cpu().tolist() == [0, 1, 2]  # the two bad edges are dropped


def test_plan_is_deterministic():
    rng = np.random.default_rng(5)
    ei = rand_edges(rng, 30000, 300, 200).to(dev())
    a = build_plan(ei, 300, 200)
    for _ in range(3):
        b = build_plan(ei, 300, 200)
        for k in ("rowptr", "col", "eid", "t_rowptr", "t_col", "t_pos"):
            assert torch.equal(a[k], b[k])


@pytest.mark.parametrize("F,ld_pad", [(6, 0), (26, 2), (64, 0), (128, 0), (306, 0), (308, 4), (512, 0), (1100, 0)])
def test_segment_mean_fwd_bwd(F, ld_pad):
    lib = _lib.require_device()
    rng = np.random.default_rng(F)
    n_src, n_dst, E = 150, 70, 900
    ei = rand_edges(rng, E, n_src, n_dst)
    ei[1, ei[1] == 3] = 4  # make row 3 empty
    p = build_plan(ei.to(dev()), n_src, n_dst)
    ld = F + ld_pad
    x_full = torch.from_numpy(rng.normal(size=(n_src, ld)).astype(np.float32))
    xg = x_full.to(dev())
    out = torch.full((n_dst, ld), 7.0, dtype=torch.float32, device=dev())
    _lib.check(lib.hmp_segment_mean_fwd(xg.data_ptr(), ld, F, p["plan"], out.data_ptr(), ld, sp()))
    x64 = x_full[:, :F].double().requires_grad_(True)
    ref = pyg_ref.scatter_mean(x64.index_select(0, ei[0]), ei[1], n_dst)
    torch.testing.assert_close(out[:, :F].cpu().double(), ref.detach(), atol=ATOL, rtol=RTOL)
    assert torch.all(out[3, :F] == 0)
    if ld_pad:
        assert torch.all(out[:, F:] == 7.0)  # padding untouched
    g_full = torch.from_numpy(rng.normal(size=(n_dst, ld)).astype(np.float32))
    gx = torch.full((n_src, ld), 7.0, dtype=torch.float32, device=dev())
    g_dev = g_full.to(dev())
    _lib.check(lib.hmp_segment_mean_bwd(g_dev.data_ptr(), ld, F, p["plan"], gx.data_ptr(), ld, sp()))
    ref.backward(g_full[:, :F].double())
    torch.testing.assert_close(gx[:, :F].cpu().double(), x64.grad, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize(
    "M,N,K,ta,tb",
    [(1, 1, 1, 0, 1), (37, 29, 6, 0, 1), (235, 128, 6, 0, 1), (2831, 192, 306, 0, 1), (2831, 306, 192, 0, 0),
     (192, 307, 2831, 1, 0), (64, 64, 64, 1, 1), (5000, 512, 256, 0, 1), (33, 65, 31, 0, 0), (40000, 256, 256, 0, 1)],
)
def test_gemm_f32(M, N, K, ta, tb):
    lib = _lib.require_device()
    rng = np.random.default_rng(M * 7 + N)
    A = torch.from_numpy(rng.uniform(-1, 1, size=(M, K)).astype(np.float32))
    B = torch.from_numpy(rng.uniform(-1, 1, size=(K, N)).astype(np.float32))
    a_mem = (A.t() if ta else A).contiguous().to(dev())
    b_mem = (B.t() if tb else B).contiguous().to(dev())
    Cd = torch.full((M, N + 3), 5.0, dtype=torch.float32, device=dev())
    _lib.check(lib.hmp_gemm_f32(a_mem.data_ptr(), a_mem.stride(0), ta, b_mem.data_ptr(), b_mem.stride(0), tb, Cd.data_ptr(),
                                N + 3, M, N, K, sp()))
    ref = A.double() @ B.double()
    # fp32 FMA chain of length K with |a*b| <= 1: error bound ~ K * 2^-24 * sqrt-ish; use the north_star tolerance
    # scaled by the magnitude of the accumulated sum
    tol = ATOL * max(1.0, float(ref.abs().max()))
    assert float((Cd[:, :N].cpu().double() - ref).abs().max()) <= tol
    assert torch.all(Cd[:, N:] == 5.0)


def test_gemm_asymmetric_operands_catch_transposition():
    """A = I, B asymmetric: a swapped accumulator row/col map would give C = B^T (guide section 3)."""
    lib = _lib.require_device()
    n = 96
    A = torch.eye(n)
    B = torch.arange(n * n, dtype=torch.float32).view(n, n)
    Cd = torch.zeros(n, n, device=dev())
    Ad, Bd = A.to(dev()), B.to(dev())  # keep the device tensors alive while the kernel runs
    _lib.check(lib.hmp_gemm_f32(Ad.data_ptr(), n, 0, Bd.data_ptr(), n, 0, Cd.data_ptr(), n, n, n, n, sp()))
    assert torch.equal(Cd.cpu(), B)
    _lib.check(lib.hmp_gemm_f32(Bd.data_ptr(), n, 0, Ad.data_ptr(), n, 1, Cd.data_ptr(), n, n, n, n, sp()))
    assert torch.equal(Cd.cpu(), B)
    _lib.check(lib.hmp_gemm_f32(Bd.data_ptr(), n, 1, Ad.data_ptr(), n, 0, Cd.data_ptr(), n, n, n, n, sp()))
    assert torch.equal(Cd.cpu(), B.t())


@pytest.mark.parametrize("n,c", [(1, 26), (235, 26), (5000, 15), (7, 3)])
def test_masked_ce(n, c):
    lib = _lib.require_device()
    rng = np.random.default_rng(n)
    logits = torch.from_numpy(rng.normal(0, 3, size=(n, c)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, c, size=n).astype(np.int64))
    ignored = c - 1
    ld = ((c + 3) // 4) * 4
    lg = torch.zeros(n, ld, device=dev())
    lg[:, :c] = logits.to(dev())
    grad = torch.full((n, ld), 9.0, device=dev())
    out2 = torch.zeros(2, device=dev())
    lab_dev = labels.to(dev())
    _lib.check(lib.hmp_masked_ce(lg.data_ptr(), ld, n, c, lab_dev.data_ptr(), ignored, grad.data_ptr(), ld, out2.data_ptr(), sp()))
    mask = labels != ignored
    l64 = logits.double().requires_grad_(True)
    if int(mask.sum()) == 0:
        assert out2.cpu().tolist() == [0.0, 0.0]
        return
    ref = torch.nn.functional.cross_entropy(l64[mask], labels[mask], reduction="sum")
    ref.backward()
    got = out2.cpu().double()
    assert int(got[1]) == int(mask.sum())
    torch.testing.assert_close(got[0], ref.detach(), atol=1e-5 * max(1, int(mask.sum())), rtol=1e-5)
    torch.testing.assert_close(grad[:, :c].cpu().double(), l64.grad, atol=ATOL, rtol=RTOL)
    assert torch.all(grad[:, c:] == 0)


def test_adam_matches_torch():
    lib = _lib.require_device()
    torch.manual_seed(0)
    n = 10007
    p0 = torch.randn(n)
    ref_p = p0.clone().double().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=0.002, weight_decay=0.001)
    p = p0.clone().to(dev())
    m = torch.zeros(n, device=dev())
    v = torch.zeros(n, device=dev())
    for step in range(1, 6):
        g = torch.randn(n)
        ref_p.grad = g.double()
        opt.step()
        g_dev = g.to(dev())
        _lib.check(lib.hmp_adam_flat(p.data_ptr(), g_dev.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.002, 0.9, 0.999,
                                     1e-8, 0.001, step, None, sp()))
        torch.cuda.synchronize()
    torch.testing.assert_close(p.cpu().double(), ref_p.detach(), atol=1e-6, rtol=1e-5)


def test_dropout_mask_rate_and_determinism():
    lib = _lib.require_device()
    n, F = 4000, 64
    a = torch.zeros(n * F, dtype=torch.uint8, device=dev())
    b = torch.zeros(n * F, dtype=torch.uint8, device=dev())
    c = torch.zeros(n * F, dtype=torch.uint8, device=dev())
    _lib.check(lib.hmp_dropout_mask(1234, 1, 3, 0.25, n, F, a.data_ptr(), sp()))
    _lib.check(lib.hmp_dropout_mask(1234, 1, 3, 0.25, n, F, b.data_ptr(), sp()))
    _lib.check(lib.hmp_dropout_mask(1234, 2, 3, 0.25, n, F, c.data_ptr(), sp()))
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    keep = float(a.float().mean())
    assert abs(keep - 0.75) < 0.005


@pytest.mark.parametrize("M,N,K,ta,tb", [(256, 256, 64, 0, 1), (1000, 300, 257, 0, 1), (513, 768, 256, 0, 0), (300, 257, 5000, 1, 0),
                                         (128, 128, 32, 1, 1), (77, 130, 33, 0, 1),
                                         (5000, 300, 260, 0, 1), (4200, 256, 768, 0, 0), (4100, 520, 96, 0, 1)])  # 256x256-tile variant
def test_gemm_bf16_matches_bf16_rounded_inputs(M, N, K, ta, tb):
    """hmp_gemm_bf16 = fp32-accumulated product of the bf16-ROUNDED operands (round to nearest even): compared with exactly that
    in float64 (tolerance = fp32 accumulation of K terms), for every operand layout and ragged sizes (edge loaders)."""
    lib = _lib.require_device()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    a, b = A.to(dev()), B.to(dev())
    c = torch.empty(M, N, device=dev())
    _lib.check(lib.hmp_gemm_bf16(a.data_ptr(), a.stride(0), ta, b.data_ptr(), b.stride(0), tb, c.data_ptr(), N, M, N, K, _lib.stream_ptr()))
    Ar = A.bfloat16().double()
    Br = B.bfloat16().double()
    ref = (Ar.t() if ta else Ar) @ (Br.t() if tb else Br)
    torch.testing.assert_close(c.cpu().double(), ref, atol=2e-4 * (K ** 0.5) / 16 + 1e-5, rtol=2e-5)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
def test_gemm_128x128_register_tiling_is_bit_identical_to_64x64(ta, tb, monkeypatch):
    """Big problems (>= 256 tiles of 128x128, >= 1e9 MACs) take the 2x2-accumulator form of gemm_kernel: the k order of every
    output element is unchanged, so the product is bit-identical to the 64x64 form (HMP_GEMM_BIG=0) -- and within fp32 round-off
    of a float64 product.  NT (x W^T), NN (dZ W) and TN (dZ^T x) operand layouts, ragged edges."""
    import ctypes as C

    from hydra_gnn_amd import _lib

    lib = _lib.require_device()
    M, N, K = 16384 + 37, 384 + 5, 306
    rng = np.random.default_rng(3)
    a = torch.from_numpy(rng.normal(size=(K, M) if ta else (M, K)).astype(np.float32)).to(dev())
    b = torch.from_numpy(rng.normal(size=(N, K) if tb else (K, N)).astype(np.float32)).to(dev())

    def run():
        c = torch.empty(M, N, dtype=torch.float32, device=dev())
        _lib.check(lib.hmp_gemm_f32(a.data_ptr(), a.stride(0), ta, b.data_ptr(), b.stride(0), tb, c.data_ptr(), c.stride(0), M, N, K,
                                    _lib.stream_ptr()))
        torch.cuda.synchronize()
        return c

    monkeypatch.setenv("HMP_GEMM_X3", "0")  # (this test compares two tile shapes of the fp32-MFMA kernel)
    big = run()
    monkeypatch.setenv("HMP_GEMM_BIG", "0")
    small = run()
    assert torch.equal(big, small)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    torch.testing.assert_close(big.double(), ref, atol=2e-4, rtol=1e-5)  # |sum of 306 products of N(0,1)| ~ 17: 1e-5 relative


@pytest.mark.parametrize("M,N,K,c16,pad", [(40000, 768, 256, True, 0), (40003, 700, 256, True, 8), (33000, 256, 128, True, 0),
                                            (36000, 768, 64, False, 8), (32768, 64, 16, True, 0), (50001, 516, 256, False, 0)])
def test_bf16_activation_projection_weight_stationary_kernel(monkeypatch, M, N, K, c16, pad):
    """hmp_gemm_bf16_a16 (C = A * W^T, A stored as bf16): tall problems run on the weight-stationary kernel (W slice in registers, A
    through swizzled LDS by direct loads, transposed product, coalesced bf16 stores); HMP_GEMM_WS=0 sends the same call to the
    tiled kernel.  Both against float64 on the rounded operands; ragged M / N, padded leading dimensions, every K class."""
    lib = _lib.require_device()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K + pad, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    W = torch.randn(N, K + pad, device="cuda", generator=g) * 0.1
    ref = A[:, :K].double() @ W[:, :K].to(torch.bfloat16).double().t()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HMP_GEMM_WS", mode)
        Cc = torch.full((M, N + pad), 7.0, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
        _lib.check(lib.hmp_gemm_bf16_a16(A.data_ptr(), K + pad, W.data_ptr(), K + pad, Cc.data_ptr(), N + pad, 1 if c16 else 0, M, N, K,
                                         _lib.stream_ptr()))
        torch.cuda.synchronize()
        got = Cc[:, :N].double()
        tol = 8e-3 if c16 else 2e-5  # bf16 output: half an ulp of 2^-8 relative; fp32: accumulation order only
        err = ((got - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        assert err < tol, (mode, err)
        if pad:
            assert torch.all(Cc[:, N:] == 7.0)  # nothing written past the N columns
        outs[mode] = Cc[:, :N].clone()
    monkeypatch.delenv("HMP_GEMM_WS")
    # same products in the same k order on the same matrix pipe: the two kernels agree to the last bit
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("M,N", [(40000, 768), (33001, 260)])
def test_fp32_activation_projection_weight_stationary_kernel(monkeypatch, M, N):
    """hmp_gemm_bf16 (fp32 A rounded to bf16 on the way into the MFMA) at K = 256, tall: the fp32-A form of the weight-stationary
    kernel (rows through LDS in two parts of 128 floats) against the tiled kernel (bit for bit) and float64 on the rounded operands"""
    lib = _lib.require_device()
    K = 256
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn(M, K, device="cuda", generator=g) * 0.5
    W = torch.randn(N, K, device="cuda", generator=g) * 0.1
    ref = A.to(torch.bfloat16).double() @ W.to(torch.bfloat16).double().t()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HMP_GEMM_WS", mode)
        Cc = torch.empty(M, N, device="cuda")
        _lib.check(lib.hmp_gemm_bf16(A.data_ptr(), K, 0, W.data_ptr(), K, 1, Cc.data_ptr(), N, M, N, K, _lib.stream_ptr()))
        torch.cuda.synchronize()
        err = ((Cc.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        assert err < 2e-5, (mode, err)
        outs[mode] = Cc
    monkeypatch.delenv("HMP_GEMM_WS")
    assert torch.equal(outs["1"], outs["0"])


# ---- round 3: operand-stationary backward GEMMs of the 10^6-node regime (csrc/gemm_bf16_bwd.hip) ---------------------------------
def _act_factor(h, act, drop_on, scale):
    """act'(h) . dropout factor from the STORED activations: a dropped element is -0 (sign bit of a zero)"""
    hf = h.float()
    dropped = (h.view(torch.int16) == -32768) if drop_on else torch.zeros_like(hf, dtype=torch.bool)
    if act == 1:
        f = (hf > 0).float() * scale
    elif act == 2:  # stored h = scale * elu(x): elu'(x) * scale = scale for x > 0, else scale * e^x = h + scale
        f = torch.where(hf > 0, torch.full_like(hf, scale), hf + scale)
    else:
        f = torch.full_like(hf, scale)
    return torch.where(dropped, torch.zeros_like(hf), f)


@pytest.mark.parametrize("M,K,split,mask", [(40037, 768, 512, "relu_drop"), (33000, 512, 256, "relu_drop"), (36000, 256, 0, "none"),
                                            (40000, 768, 0, "relu"), (34567, 512, 0, "elu_drop")])
def test_bf16_input_gradient_weight_stationary_kernel(monkeypatch, M, K, split, mask):
    """hmp_gemm_bf16_dx: G = act'(H) . (dZ * Wp) with dZ in one or two bf16 pieces, against float64 on the rounded operands and
    against the tiled kernel (HMP_GEMM_DX=0) on the same call: every K class, ragged M, padded leading dimensions, the
    activation / dropout mask read from the stored bf16 activations."""
    lib = _lib.require_device()
    N, pad = 256, 8
    g = torch.Generator(device="cuda").manual_seed(M + K)
    k1 = split if split else K
    A1 = (torch.randn(M, k1 + pad, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    A2 = (torch.randn(M, (K - k1) + pad, device="cuda", generator=g) * 0.5).to(torch.bfloat16) if split else None
    W = torch.randn(K, N + 4, device="cuda", generator=g) * 0.1
    act = {"none": 0, "relu": 1, "relu_drop": 1, "elu_drop": 2}[mask]
    drop_on = int(mask.endswith("drop"))
    scale = 1.0 / 0.75 if drop_on else 1.0
    H = None
    if mask != "none":
        H = (torch.randn(M, N + pad, device="cuda", generator=g)).to(torch.bfloat16)
        if drop_on:
            dropm = torch.rand(M, N + pad, device="cuda", generator=g) < 0.25
            H = torch.where(dropm, torch.tensor(-0.0, device="cuda", dtype=torch.bfloat16), H)
            assert int((H.view(torch.int16) == -32768).sum()) > M  # the -0 pattern survives
    Afull = torch.cat([A1[:, :k1], A2[:, : K - k1]], dim=1) if split else A1[:, :K]
    ref = Afull.double() @ W[:, :N].to(torch.bfloat16).double()
    if H is not None:
        ref = ref * _act_factor(H[:, :N], act, drop_on, scale).double()
    outs = {}
    for mode in ("2", "0"):  # 2: the weight-stationary kernel for every K class (K = 768 is off by default: slower than the tiled one)
        monkeypatch.setenv("HMP_GEMM_DX", mode)
        G = torch.full((M, N + pad), 7.0, device="cuda", dtype=torch.bfloat16)
        _lib.check(lib.hmp_gemm_bf16_dx(A1.data_ptr(), k1 + pad, A2.data_ptr() if split else None, (K - k1) + pad if split else 0, split,
                                        W.data_ptr(), N + 4, H.data_ptr() if H is not None else None, N + pad, act, drop_on, scale,
                                        G.data_ptr(), N + pad, M, N, K, _lib.stream_ptr()))
        torch.cuda.synchronize()
        err = ((G[:, :N].double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        assert err < 8e-3, (mode, err)  # bf16 output: half an ulp of 2^-8 relative
        assert torch.all(G[:, N:] == 7.0)
        outs[mode] = G[:, :N].clone()
    monkeypatch.delenv("HMP_GEMM_DX")
    # same products, same k order, fp32 accumulation on the same matrix pipe: the bf16 results agree (a last-bit difference in the
    # fp32 sum can move a bf16 rounding: allow a handful)
    diff = (outs["2"].float() - outs["0"].float()).abs() > 0
    assert int(diff.sum()) <= M * N // 2000, int(diff.sum())


@pytest.mark.parametrize("nodes,Mw,split,h_bf16", [(70001, 768, 512, 1), (66000, 512, 256, 1), (80000, 768, 512, 0), (65536, 256, 0, 1)])
def test_bf16_weight_gradient_output_stationary_kernel(monkeypatch, nodes, Mw, split, h_bf16):
    """hmp_gemm_bf16_dw: slabs of dW = dZ^T * [H | 1] over node ranges; their sum against float64 on the rounded operands and against
    the tiled kernel (HMP_GEMM_DW=0): bf16 and fp32 H, dZ in two pieces, a node count that leaves a partial last chunk."""
    import ctypes as C

    lib = _lib.require_device()
    F, pad, ldc = 256, 8, 260
    g = torch.Generator(device="cuda").manual_seed(nodes + Mw)
    k1 = split if split else Mw
    A1 = (torch.randn(nodes, k1 + pad, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    A2 = (torch.randn(nodes, (Mw - k1) + pad, device="cuda", generator=g) * 0.5).to(torch.bfloat16) if split else None
    Hf = torch.randn(nodes, F + pad, device="cuda", generator=g)
    Hm = Hf.to(torch.bfloat16) if h_bf16 else Hf
    Afull = torch.cat([A1[:, :k1], A2[:, : Mw - k1]], dim=1) if split else A1[:, :Mw]
    Hr = Hm[:, :F].to(torch.bfloat16).double()
    ref = torch.cat([Afull.double().t() @ Hr, Afull.double().sum(0, keepdim=True).t()], dim=1)  # [Mw, F + 1]
    sums = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HMP_GEMM_DW", mode)
        max_slabs = 192
        slabs = torch.full((max_slabs, Mw, ldc), float("nan"), device="cuda")
        ns = C.c_int32(0)
        _lib.check(lib.hmp_gemm_bf16_dw(A1.data_ptr(), k1 + pad, A2.data_ptr() if split else None, (Mw - k1) + pad if split else 0, split,
                                        Hm.data_ptr(), F + pad, h_bf16, slabs.data_ptr(), ldc, Mw * ldc, max_slabs, C.byref(ns), Mw, F, nodes,
                                        _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert 1 <= ns.value <= max_slabs
        got = slabs[: ns.value, :, : F + 1].double().sum(0)
        assert torch.isfinite(got).all(), mode
        err = ((got - ref).abs() / ref.abs().clamp_min(10.0)).max().item()
        assert err < 2e-4, (mode, err, ns.value)
        sums[mode] = got
    monkeypatch.delenv("HMP_GEMM_DW")
    assert ((sums["1"] - sums["0"]).abs() / ref.abs().clamp_min(10.0)).max().item() < 2e-4


@pytest.mark.parametrize("shape", [(70001, 192, 306, 1, 306, 306), (40000, 192, 64, 0, 64, 192), (33000, 130, 80, 1, 84, 84)])
def test_fp32_gemm_whole_width_tiles(shape, monkeypatch):
    """Outputs of 129 .. 192 columns over >= 10^9 multiply-adds (an MP3D layer at the reference's batch size: 190 k x 192) run on
    64 x 192 tiles -- one workgroup per 64 rows, three accumulator tiles per wave, the B image 192 columns wide (not a power of
    two: the rotated LDS image wraps by comparison).  The k order of every output element is unchanged, so the product is
    bit-identical to the 64x64 form (HMP_GEMM_WIDE=0) and within fp32 round-off of float64; row pitch 306 = 8-byte rows."""
    from hydra_gnn_amd import _lib

    lib = _lib.require_device()
    M, N, K, tb, lda, ldb = shape
    torch.manual_seed(6)
    A = torch.randn(M, lda, device=dev())
    B = torch.randn(N, ldb, device=dev()) if tb else torch.randn(K, ldb, device=dev())

    def run():
        C = torch.full((M, N + 3), float("nan"), device=dev())
        _lib.check(lib.hmp_gemm_f32(A.data_ptr(), lda, 0, B.data_ptr(), ldb, tb, C.data_ptr(), N + 3, M, N, K, _lib.stream_ptr()))
        torch.cuda.synchronize()
        return C

    monkeypatch.setenv("HMP_GEMM_X3", "0")  # (this test compares two tile shapes of the fp32-MFMA kernel)
    C1 = run()
    monkeypatch.setenv("HMP_GEMM_WIDE", "0")
    C0 = run()
    monkeypatch.delenv("HMP_GEMM_WIDE")
    assert torch.isnan(C1[:, N:]).all() and not torch.isnan(C1[:, :N]).any()
    assert torch.equal(C1[:, :N], C0[:, :N])
    ref = A[:, :K].double() @ (B[:, :K].double().t() if tb else B[:K, :N].double())
    assert (C1[:, :N].double() - ref).abs().max().item() < 1e-4 * max(K, 64) ** 0.5


@pytest.mark.parametrize("ta,tb,M,N,K,lda_pad,ldb_pad", [
    (0, 1, 70001, 192, 306, 0, 0),     # NT, the batch-2048 layer-0 projection: rows of 306 floats (8-byte rows), ragged M
    (0, 1, 16384 + 37, 389, 306, 0, 0),  # NT, ragged everywhere
    (0, 1, 9000, 515, 307, 0, 0),      # NT, odd pitch: element loads
    (0, 0, 90000, 64, 192, 0, 0),      # NN (input gradient), N below a tile
    (0, 0, 20011, 300, 260, 2, 2),     # NN, row-contiguous B with a partial column tile, padded pitches
    (1, 0, 192, 306, 190000, 0, 0),    # TN (weight gradient): row-contiguous operands, B with pitch 306
    (1, 0, 515, 131, 40000, 1, 3),     # TN, odd pitches
])
def test_fp32_gemm_on_the_bf16_pipe_by_three_way_split(ta, tb, M, N, K, lda_pad, ldb_pad, monkeypatch):
    """Launches of >= 10^9 multiply-adds run hmp_gemm_f32 on gemm_x3_kernel: operands split exactly into three bf16 pieces, six piece
    products per element product on v_mfma_f32_32x32x16_bf16.  Accuracy bar = the fp32-MFMA kernel's own: both within 2e-6 x the
    product's scale of the float64 product (|error| of an fp32 dot product of K terms), and the split form not worse than 2x the
    fp32-MFMA form's error.  NT / NN / TN operand layouts, ragged tiles, 16-byte / 8-byte / unaligned rows; columns past N untouched."""
    from hydra_gnn_amd import _lib

    lib = _lib.require_device()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    ar, ac = (K, M) if ta else (M, K)
    br, bc = (N, K) if tb else (K, N)
    A = torch.randn(ar, ac + lda_pad, device=dev(), generator=g)
    B = torch.randn(br, bc + ldb_pad, device=dev(), generator=g) * 0.1
    a64 = A[:, :ac].double()
    b64 = B[:, :bc].double()
    ref = (a64.t() if ta else a64) @ (b64.t() if tb else b64)

    def run():
        C = torch.full((M, N + 5), float("nan"), device=dev())
        _lib.check(lib.hmp_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, C.data_ptr(), C.stride(0), M, N, K,
                                    _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isnan(C[:, N:]).all() and not torch.isnan(C[:, :N]).any()
        return C[:, :N].double()

    monkeypatch.setenv("HMP_GEMM_X3", "0")
    f32 = run()
    e32 = (f32 - ref).abs().max().item()
    scale = (a64.abs().mean() * b64.abs().mean() * K).item()  # ~ sum of |products|
    monkeypatch.setenv("HMP_GEMM_X3", "2")  # (2: every launch form, also the split-K one the executor keeps on the tall kernel)
    for tile in ("128", "64"):  # the square tile and the 128 x 64 one (three workgroups per CU; chosen when it fills the chip better)
        monkeypatch.setenv("HMP_GEMM_X3_TILE", tile)
        x3 = run()
        assert not torch.equal(x3, f32)  # the split kernel really ran
        e3 = (x3 - ref).abs().max().item()
        assert e3 <= 2e-6 * scale, (tile, e3, scale)
        assert e3 <= 2.0 * e32 + 1e-7 * scale, (tile, e3, e32)
