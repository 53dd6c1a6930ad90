"""GPU parity of the rpg_* family, through the C ABI, against the oracle on the same Philox
coordinates.  Tolerance: the GPU and the oracle run the same fp64 algorithm on the same uniforms but
with different libm implementations (ocml vs glibc, <= 1-2 ulp each) and FMA contraction, so draws
agree to RTOL = 1e-10 relative; an accept/reject decision that lands within an ulp of its threshold can
flip (probability ~1e-15 per test), which MAX_FLIP_FRAC tolerates."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
from scipy import stats

import pgmath

pytestmark = pytest.mark.gpu
RTOL = 1e-10
MAX_FLIP_FRAC = 1e-5
HERE = os.path.dirname(os.path.abspath(__file__))


def agree(got, ref, rtol=RTOL):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    bad = np.sum(~((rel <= rtol) | (got == ref)))
    assert bad <= MAX_FLIP_FRAC * got.size, (bad, rel.max())


def dev_t(a, gpu, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=gpu)


def test_dot_C_entry_points_match_oracle(gpu, oracle):
    """The ten-symbol boundary with host buffers, exactly as R's .C calls it (LogitWrapper.R:29..118)."""
    import bayeslogit_amd as bl
    rng = np.random.default_rng(1)
    num = 3000
    z = rng.normal(0, 2, num)
    bl.set_seed(1234)
    x = bl.rpg_devroye(num, n=[1, 2, 3, 0], z=z)                     # epoch 0
    agree(x, oracle.rpg_devroye(num, np.resize([1, 2, 3, 0], num), z, 1234, 0))
    x = bl.rpg_alt(num, h=[1.0, 2.5, 4.0, 7.7, 12.0], z=z)           # epoch 1
    agree(x, oracle.rpg_alt(num, np.resize([1.0, 2.5, 4.0, 7.7, 12.0], num), z, 1234, 1))
    r = bl.rpg_sp(num, h=[14.0, 30.0, 170.0], z=z, track_iter=True)  # epoch 2
    xo, ito = oracle.rpg_sp(num, np.resize([14.0, 30.0, 170.0], num), z, 1234, 2)
    agree(r["samp"], xo)
    assert np.mean(r["iter"] == ito) > 1 - MAX_FLIP_FRAC
    x = bl.rpg_gamma(400, h=[0.3, 1.0, 2.2], z=z[:400], trunc=50)    # epoch 3
    agree(x, oracle.rpg_gamma(400, np.resize([0.3, 1.0, 2.2], 400), z[:400], 1234, 50, 3), 1e-9)
    h = np.concatenate([rng.integers(1, 51, num - 600).astype(float), rng.uniform(0.05, 0.99, 100),
                        rng.uniform(1, 13, 200), rng.uniform(13, 170, 200), rng.uniform(170, 500, 100)])
    x = bl.rpg(num, h=h, z=z)                                        # epoch 4
    agree(x, oracle.rpg_hybrid(num, h, z, 1234, 4), 1e-9)
    assert bl._lib.lib().bl_get_epoch() == 5
    # same seed again -> same sequence of calls reproduces (R: set.seed)
    bl.set_seed(1234)
    a = bl.rpg(100, 1.0, 0.5)
    bl.set_seed(1234)
    assert np.array_equal(a, bl.rpg(100, 1.0, 0.5))


def test_C1_plumbing(gpu):
    """BASELINE config C1 through the boundary: rpg(num=1e3, h=1, z=0), KS vs exact PG(1,0)."""
    import bayeslogit_amd as bl
    bl.set_seed(2024)
    x = bl.rpg(1000, 1, 0.0)
    assert x.shape == (1000,)
    assert stats.kstest(x, lambda w: pgmath.pg1_cdf(w, 0.0)).pvalue > 0.01
    assert abs(x.mean() - 0.25) < 4 * np.sqrt(1 / 24 / 1000)


@pytest.mark.parametrize("case", ["devroye", "alt", "sp", "hybrid", "thresholds"])
def test_device_entry_points_match_oracle(gpu, oracle, case):
    from bayeslogit_amd import device as D
    rng = np.random.default_rng(7)
    n = 60000
    z = np.concatenate([rng.uniform(0, 4, n // 2), rng.normal(0, 3, n - n // 2 - 4), [0.0, -0.0, 60.0, 1e-9]])
    zt = dev_t(z, gpu)
    if case == "devroye":
        nv = rng.integers(0, 5, n).astype(np.int32)
        x = D.rpg_devroye(zt, dev_t(nv, gpu, torch.int32), seed=5, epoch=2, idx0=2 ** 40)
        D.sync_status()
        agree(x.cpu().numpy(), oracle.rpg_devroye(n, nv, z, 5, 2, 2 ** 40))
    elif case == "alt":
        h = rng.uniform(1, 13, n)
        h[:100] = np.repeat([1.0, 4.0, 4.000001, 5.0, 8.0, 9.0, 12.999, 3.99, 2.0, 1.009], 10)
        x = D.rpg_alt(dev_t(h, gpu), zt, seed=6)
        D.sync_status()
        agree(x.cpu().numpy(), oracle.rpg_alt(n, h, z, 6))
    elif case == "sp":
        h = rng.uniform(13, 170, n)
        h[:6] = [13.0001, 14.0, 50.0, 100.0, 169.9, 170.0]
        it = torch.full((n,), -1, dtype=torch.int32, device=gpu)
        x = D.rpg_sp(dev_t(h, gpu), zt, seed=7, iters=it)
        D.sync_status()
        xo, ito = oracle.rpg_sp(n, h, z, 7)
        agree(x.cpu().numpy(), xo)
        assert np.mean(it.cpu().numpy() == ito) > 1 - MAX_FLIP_FRAC
    elif case == "hybrid":
        h = 1.0 + rng.integers(0, 50, n)
        x = D.rpg_hybrid(dev_t(h, gpu), zt, seed=8, epoch=1, idx0=17)
        D.sync_status()
        agree(x.cpu().numpy(), oracle.rpg_hybrid(n, h, z, 8, 1, 17), 1e-9)
    else:
        # every branch boundary of LogitWrapper.cpp:142-161
        hb = np.array([0.0, -1.0, 1e-3, 0.999, 1.0, 1.0000001, 2.0, 2.5, 13.0, 13.0000001, 170.0, 170.0000001, 1e4])
        h = np.resize(hb, 1300)
        zz = np.resize(z, 1300)
        x = D.rpg_hybrid(dev_t(h, gpu), dev_t(zz, gpu), seed=9)
        D.sync_status()
        xo = oracle.rpg_hybrid(1300, h, zz, 9)
        agree(x.cpu().numpy(), xo, 1e-9)
        assert np.all(x.cpu().numpy()[h <= 0] == 0)


def test_golden_vectors_on_gpu(gpu):
    from bayeslogit_amd import device as D
    g = json.load(open(os.path.join(HERE, "golden", "pg_golden_v2.json")))
    for d in g["hybrid_draws"]:
        k = len(d["x"])
        h = torch.full((k,), d["b"], dtype=torch.float64, device=gpu)
        z = torch.full((k,), d["z"], dtype=torch.float64, device=gpu)
        x = D.rpg_hybrid(h, z, seed=d["seed"], epoch=d["epoch"], idx0=d["idx0"])
        D.sync_status()
        agree(x.cpu().numpy(), np.array(d["x"]), 1e-9)
    d = g["devroye_n"]
    x = D.rpg_devroye(dev_t(d["z"], gpu), dev_t(d["n"], gpu, torch.int32), seed=d["seed"])
    D.sync_status()
    agree(x.cpu().numpy(), np.array(d["x"]))
    d = g["sp_iter"]
    it = torch.zeros(6, dtype=torch.int32, device=gpu)
    x = D.rpg_sp(dev_t(d["h"], gpu), torch.full((6,), d["z"], dtype=torch.float64, device=gpu), seed=d["seed"], iters=it)
    D.sync_status()
    agree(x.cpu().numpy(), np.array(d["x"]))
    assert it.cpu().tolist() == d["iter"]
    d = g["alt"]
    x = D.rpg_alt(dev_t(d["h"], gpu), torch.full((10,), d["z"], dtype=torch.float64, device=gpu), seed=d["seed"])
    D.sync_status()
    agree(x.cpu().numpy(), np.array(d["x"]))


def test_edge_cases(gpu, oracle):
    import bayeslogit_amd as bl
    from bayeslogit_amd import device as D
    # empty input
    e = torch.empty(0, dtype=torch.float64, device=gpu)
    assert D.rpg_devroye(e, 1, seed=1).numel() == 0
    assert D.rpg_hybrid(e, e, seed=1).numel() == 0
    assert bl.rpg_devroye(0).shape == (0,)
    # single element, ragged (not a multiple of 64 / 256) lengths
    for n in (1, 63, 65, 257, 1000):
        z = np.linspace(-3, 3, n)
        x = D.rpg_devroye(dev_t(z, gpu), 1, seed=3, idx0=5)
        agree(x.cpu().numpy(), oracle.rpg_devroye(n, 1, z, 3, 0, 5))
    # n < 1 is clamped to 1 and flagged (NTHROW build, PolyaGamma.cpp:128-135)
    x = D.rpg_devroye(dev_t([1.0, 1.0], gpu), dev_t([-2, 1], gpu, torch.int32), seed=4)
    with pytest.raises(bl.BayesLogitError):
        D.sync_status()
    assert bl._lib.lib().bl_last_sampler_flags() & 2
    agree(x.cpu().numpy(), oracle.rpg_devroye(2, [-2, 1], [1.0, 1.0], 4))
    D.sync_status()   # flags were cleared by the failed sync
    # shapes below 1 through rpg_alt / rpg_sp: refused and flagged, 0 returned (PolyaGammaAlt.cpp:207-210); h = 0 -> 0
    hb = dev_t([0.5, 2.0, 0.0, -3.0, 7.0], gpu)
    zb = dev_t([1.0, 1.0, 1.0, 1.0, 1.0], gpu)
    for fn, href in ((D.rpg_alt, [0.0, 2.0, 0.0, 0.0, 7.0]), (D.rpg_sp, [0.0, 2.0, 0.0, 0.0, 7.0])):
        x = fn(hb, zb, seed=4)
        with pytest.raises(bl.BayesLogitError):
            D.sync_status()
        assert bl._lib.lib().bl_last_sampler_flags() & 2
        xg = x.cpu().numpy()
        assert xg[0] == 0 and xg[2] == 0 and xg[3] == 0 and xg[1] > 0 and xg[4] > 0
        ref = oracle.rpg_alt(5, href, 1.0, 4) if fn is D.rpg_alt else oracle.rpg_sp(5, href, 1.0, 4)[0]
        agree(xg, ref)
    # task-queue chunk boundaries (512 observations per wave chunk, 64 tasks per set-up batch)
    for n in (63, 64, 65, 511, 512, 513, 2049, 5000):
        hh = rng_h = np.random.default_rng(n).integers(3, 51, n).astype(float)
        zz = np.random.default_rng(n + 1).normal(0, 1.5, n)
        x = D.rpg_hybrid(dev_t(hh, gpu), dev_t(zz, gpu), seed=10, idx0=7)
        D.sync_status()
        agree(x.cpu().numpy(), oracle.rpg_hybrid(n, hh, zz, 10, 0, 7), 1e-9)
    # chunk boundaries of the per-class work queue (256 observations per wave chunk), both classes mixed,
    # shapes > 1, huge |z| (proposal mass underflows to 0), z = NaN (the reference falls through with NaN)
    rng = np.random.default_rng(8)
    for n in (255, 256, 257, 511, 512, 513, 2047, 4100):
        z = rng.normal(0, 3, n)
        z[::97] = 60.0
        z[5] = np.nan
        shp = rng.integers(1, 4, n).astype(np.int32)
        x = D.rpg_devroye(dev_t(z, gpu), dev_t(shp, gpu, torch.int32), seed=9, idx0=123456789012)
        D.sync_status()
        xo = oracle.rpg_devroye(n, shp, z, 9, 0, 123456789012)
        xg = x.cpu().numpy()
        assert np.isnan(xg[5]) and np.isnan(xo[5])
        keep = np.arange(n) != 5
        agree(xg[keep], xo[keep])


def test_determinism_and_shard_invariance(gpu):
    """Fixed counter seed => identical output; an index range split over ranks gives the same draws."""
    from bayeslogit_amd import device as D
    n = 300001
    z = torch.empty(n, dtype=torch.float64, device=gpu)
    D.fill_unif(z, 0.0, 4.0, 20240001)
    a = D.rpg_devroye(z, 1, seed=77, epoch=5)
    b = D.rpg_devroye(z, 1, seed=77, epoch=5)
    assert torch.equal(a, b)
    cut = 123457
    lo = D.rpg_devroye(z[:cut].contiguous(), 1, seed=77, epoch=5, idx0=0)
    hi = D.rpg_devroye(z[cut:].contiguous(), 1, seed=77, epoch=5, idx0=cut)
    assert torch.equal(torch.cat([lo, hi]), a)
    c = D.rpg_devroye(z, 1, seed=77, epoch=6)
    assert not torch.equal(a, c)
    h = torch.empty(n, dtype=torch.float64, device=gpu)
    D.fill_shape(h, 50, 20240001, epoch=1)
    assert int(h.min()) == 1 and int(h.max()) == 50
    m1 = D.rpg_hybrid(h, z, seed=78)
    m2 = torch.cat([D.rpg_hybrid(h[:cut].contiguous(), z[:cut].contiguous(), seed=78),
                    D.rpg_hybrid(h[cut:].contiguous(), z[cut:].contiguous(), seed=78, idx0=cut)])
    assert torch.equal(m1, m2)
    D.sync_status()


def test_full_size_C2_properties(gpu):
    """BASELINE C2 at full size (1e8 draws): size-independent properties -- positivity, sample moments
    against the closed forms E[PG(1,z)] = tanh(z/2)/(2z) (the reference tests' criterion) averaged over
    z ~ U(0,4), KS of a z-slice against the exact CDF, and a checksum that is stable across two runs."""
    from bayeslogit_amd import device as D
    n = 100_000_000
    z = torch.empty(n, dtype=torch.float64, device=gpu)
    D.fill_unif(z, 0.0, 4.0, 20240001)
    x = D.rpg_devroye(z, 1, seed=20240002)
    D.sync_status()
    assert bool((x > 0).all()) and bool(torch.isfinite(x).all())
    m_expect = (torch.tanh(z / 2) / (2 * z)).mean().item()
    v_expect = ((torch.sinh(z) - z) / (4 * z ** 3 * torch.cosh(z / 2) ** 2)).mean().item()
    assert abs(x.mean().item() - m_expect) < 6 * np.sqrt(0.05 / n)
    resid = x - torch.tanh(z / 2) / (2 * z)
    assert abs((resid ** 2).mean().item() - v_expect) < 1e-5
    sel = (z > 1.99) & (z < 2.01)
    xs = x[sel][:200000].cpu().numpy()
    assert stats.kstest(xs, lambda w: pgmath.pg1_cdf(w, 2.0)).pvalue > 1e-4
    s1 = x.sum().item()
    x2 = D.rpg_devroye(z, 1, seed=20240002)
    assert torch.equal(x, x2) and x2.sum().item() == s1


def test_every_element_is_written_once_without_a_zeroing_launch(gpu, oracle):
    """rpg_hybrid / rpg_alt / rpg_sp no longer zero x first: every element is written by exactly one kernel -- members by
    their class's pass, the b <= 0 branch (LogitWrapper.cpp:159-161) by the first pass's scan, h == 0 and refused shapes
    (rpg_alt / rpg_sp) by the scan, a two-task observation's zero by the scan before its tasks add.  Output buffers
    pre-filled with NaN: none may survive, and the values are the oracle's; the class passes launched one by one
    (bl_diag_rpg_hybrid_class_dev) write exactly their own members."""
    import bayeslogit_amd as bl
    from bayeslogit_amd import device as D
    rng = np.random.default_rng(21)
    n = 70001
    h = rng.integers(1, 51, n).astype(float)
    h[::13] = 0.0
    h[1::29] = -2.0
    h[2::31] = 0.37                      # sum of gammas
    h[3::37] = 200.0                     # normal approximation
    h[4::41] = 7.5
    z = rng.normal(0, 1.5, n)
    ht, zt = dev_t(h, gpu), dev_t(z, gpu)
    x = torch.full((n,), float("nan"), dtype=torch.float64, device=gpu)
    D.rpg_hybrid(ht, zt, seed=77, idx0=3, out=x)
    D.sync_status()
    xg = x.cpu().numpy()
    assert not np.isnan(xg).any()
    agree(xg, oracle.rpg_hybrid(n, h, z, 77, 0, 3), 1e-9)
    assert np.all(xg[h <= 0] == 0.0)
    # one pass at a time: its own class and nothing else
    cls_of = np.where(h > 170, 5, np.where(h > 13, 4, np.where((h == 1) | (h == 2), 2, np.where(h > 1, 3, np.where(h > 0, 1, 0)))))
    for cls in (1, 2, 3, 4, 5):
        y = torch.full((n,), float("nan"), dtype=torch.float64, device=gpu)
        D.rpg_hybrid_class(ht, zt, cls, seed=77, idx0=3, out=y)
        D.sync_status()
        yg = y.cpu().numpy()
        assert np.array_equal(~np.isnan(yg), cls_of == cls), cls
        assert np.array_equal(yg[cls_of == cls], xg[cls_of == cls])
    # rpg_alt / rpg_sp: h == 0 -> 0, a shape below 1 refused (flagged) -> 0, everything written
    ha = rng.uniform(1.0, 40.0, n)
    ha[::17] = 0.0
    ha[5::19] = 0.5
    for fn, ofn in ((D.rpg_alt, oracle.rpg_alt), (D.rpg_sp, lambda *a_: oracle.rpg_sp(*a_)[0])):
        y = torch.full((n,), float("nan"), dtype=torch.float64, device=gpu)
        fn(dev_t(ha, gpu), zt, seed=78, idx0=9, out=y)
        with pytest.raises(bl.BayesLogitError):
            D.sync_status()                          # the refused shapes raise BL_ST_BAD_SHAPE
        yg = y.cpu().numpy()
        assert not np.isnan(yg).any() and np.all(yg[ha < 1.0] == 0.0)
        href = np.where(ha < 1.0, 0.0, ha)
        agree(yg, ofn(n, href, z, 78, 0, 9), 1e-9)


def test_work_counts_of_the_draws(gpu):
    """bl_diag_count_blocks_dev (bench.py's attempts per draw): an exact replay of the streams.  Devroye: every draw needs at
    least one block and 1.1-1.6 on average at z ~ U(0,4); the saddle-point class's blocks are at least the iteration counts
    rpg_sp returns (PolyaGammaSP::draw's `iter` counts its outer loop, an attempt is any pass of the body); class sizes
    are those of rpg_hybrid's dispatch."""
    from bayeslogit_amd import device as D
    n = 200_000
    z = torch.empty(n, dtype=torch.float64, device=gpu)
    D.fill_unif(z, 0.0, 4.0, 5)
    c = D.count_blocks(None, z, seed=6)
    d = c["devroye"]
    assert d["observations"] == n and d["draws"] == n and 1.1 * n < d["blocks"] < 1.6 * n
    wz = c["devroye_wide_z"]
    assert abs(wz["observations"] / n - (4.0 - 3.125) / 4.0) < 0.01           # |z|/2 >= 1/0.64
    h = torch.empty(n, dtype=torch.float64, device=gpu)
    D.fill_shape(h, 50, 5, epoch=1)
    c = D.count_blocks(h, z, seed=6)
    hh = h.cpu().numpy()
    assert c["saddle_point"]["observations"] == int((hh > 13).sum())
    assert c["alternating_series"]["observations"] == int(((hh > 2) & (hh <= 13)).sum())
    assert c["devroye"]["observations"] == int((hh <= 2).sum()) and c["devroye"]["draws"] == int(hh[hh <= 2].sum())
    it = torch.zeros(n, dtype=torch.int32, device=gpu)
    sp = h > 13
    D.rpg_sp(h[sp].contiguous(), z[sp].contiguous(), seed=6, iters=it[: int(sp.sum())])
    D.sync_status()
    # (the same shapes and z on other stream indices: compared in the mean, not row by row)
    mean_iter = it[: int(sp.sum())].double().mean().item()
    apd = c["saddle_point"]["blocks"] / c["saddle_point"]["draws"]
    assert 1.0 <= mean_iter <= apd < 2.2, (mean_iter, apd)


def test_full_size_C3_properties(gpu):
    """BASELINE C3 (b in 1..50, z ~ N(0, sd^2 = 2)) at its full 1e8 draws: per-shape sample mean vs pg_m1."""
    from bayeslogit_amd import device as D
    n = 100_000_000
    z = torch.empty(n, dtype=torch.float64, device=gpu)
    h = torch.empty(n, dtype=torch.float64, device=gpu)
    D.fill_norm(z, 0.0, 2 ** 0.5, 20240001)
    D.fill_shape(h, 50, 20240001, epoch=1)
    x = D.rpg_hybrid(h, z, seed=20240002)
    D.sync_status()
    assert bool((x > 0).all())
    za = z.abs().clamp_min(1e-12)
    mean_i = h * torch.tanh(za / 2) / (2 * za)
    var_i = h * (torch.sinh(za) - za) / (4 * za ** 3 * torch.cosh(za / 2) ** 2)
    for b in (1, 2, 3, 7, 13, 14, 30, 50):
        m = h == b
        k = int(m.sum())
        err = (x[m] - mean_i[m]).mean().item()
        assert abs(err) < 6 * np.sqrt(var_i[m].mean().item() / k), b


def test_matrix_pipe_diagnostic(gpu):
    """bl_diag_mfma_f64_dev (the rate bench.py quotes beside the X'Omega X pass): a plausible fp64 rate, and
    the argument checks."""
    from bayeslogit_amd import _lib, device as D
    tf = D.mfma_f64_sustained_tflops(2, iters=500)
    assert 5.0 < tf < 200.0
    fl = C.c_double(0.0)
    work = torch.empty(64, dtype=torch.float64, device=gpu)
    rc = _lib.lib().bl_diag_mfma_f64_dev(work.data_ptr(), work.numel(), 2, 10, C.byref(fl), None)
    assert rc != 0          # work buffer too small
    rc = _lib.lib().bl_diag_mfma_f64_dev(work.data_ptr(), work.numel(), 0, 10, C.byref(fl), None)
    assert rc != 0
