"""include/dvbs2gpu_math.h -- the engine's own sin/cos/atan2/exp/log (the reference calls libm there: freq_shift.cpp:6,
dvbs2_pll.cpp:39,50-75, dvbs2_plhdr_demod.cpp:35, fll.cpp:137, constellation.cpp:226,250,259).

CPU part: the host evaluation is accurate (so the receiver built on it behaves like one built on libm).
GPU part: gfx950 evaluates the same header to the SAME BITS as x86-64 -- the property every bit-exact float-stage test
of this repository rests on."""
import numpy as np
import pytest
import orc


def ulp_err(got, want64):
    """error of float32 `got` against float64 `want64` in units of the float32 spacing at want64"""
    w32 = want64.astype(np.float32)
    sp = np.spacing(np.abs(w32)).astype(np.float64)
    sp = np.maximum(sp, np.float64(np.finfo(np.float32).tiny) * 2.0 ** -23)
    return np.abs(got.astype(np.float64) - want64) / sp


def inputs(rng, n):
    return {
        'phase': np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, n), rng.uniform(-70, 70, n // 4), [0.0, -0.0, np.pi, -np.pi, 2 * np.pi,
                                                                                                      -2 * np.pi, np.pi / 4, 1e-30, -1e-30]]).astype(np.float32),
        'xy': (np.concatenate([rng.standard_normal(n), rng.standard_normal(n // 4) * 1e-20, rng.standard_normal(n // 4) * 1e15,
                               [0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 1.0, 1.0]]).astype(np.float32),
               np.concatenate([rng.standard_normal(n), rng.standard_normal(n // 4) * 1e-20, rng.standard_normal(n // 4) * 1e15,
                               [0.0, 1.0, 0.0, 0.0, -1.0, -1.0, 1.0, -1.0]]).astype(np.float32)),
        'exp': np.concatenate([-rng.uniform(0, 110, n), rng.uniform(-1, 1, n // 4), rng.uniform(80, 95, 64),
                               [0.0, -0.0, -103.9, -104.5, -87.3, -88.0, 88.7, 89.5, -1e-30]]).astype(np.float32),
        'log': np.concatenate([np.exp(rng.uniform(-103, 88, n)), rng.uniform(0.5, 2.0, n // 4), [0.0, 1.0, 1e-45, 3e-39, 3.4e38,
                                                                                               np.inf, 1.4142135, 1.4142137, 0.70710677]]).astype(np.float32),
        'clamp': np.concatenate([rng.standard_normal(n) * 300, [np.inf, -np.inf, np.nan, 127.0, 127.5, -127.0, -127.5, 254.1, 1e30, 0.99]]).astype(np.float32),
    }


def test_host_accuracy():
    rng = np.random.default_rng(1)
    I = inputs(rng, 200000)
    ph = I['phase']
    sn, cs = orc.math_eval(0, ph)
    # error measured against the magnitude scale of the results (|sin|, |cos| <= 1): absolute error in units of 2^-24
    assert np.max(np.abs(sn - np.sin(ph.astype(np.float64)))) < 2.5 * 2.0 ** -24
    assert np.max(np.abs(cs - np.cos(ph.astype(np.float64)))) < 2.5 * 2.0 ** -24
    z = np.zeros(1, np.float32)
    s0, c0 = orc.math_eval(0, z)
    assert s0[0] == 0.0 and c0[0] == 1.0
    y, x = I['xy']
    fin = np.isfinite(y.astype(np.float64) / np.maximum(np.abs(x.astype(np.float64)), 1e-300))
    at, _ = orc.math_eval(1, y, x)
    assert np.max(ulp_err(at[fin], np.arctan2(y.astype(np.float64), x.astype(np.float64))[fin])) < 3.0
    for yy, xx, want in [(0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, np.pi / 2), (0.0, -1.0, np.pi), (-1.0, 0.0, -np.pi / 2)]:
        g, _ = orc.math_eval(1, np.float32([yy]), np.float32([xx]))
        assert g[0] == np.float32(want), (yy, xx, g[0])
    e = I['exp']
    ex, _ = orc.math_eval(2, e)
    with np.errstate(over='ignore'):
        want = np.exp(e.astype(np.float64))
    ok = (e > -103.0) & (e < 88.0)
    assert np.max(ulp_err(ex[ok], want[ok])) <= 0.5 + 1e-6          # correctly rounded up to double rounding
    assert ex[e < -104.0].max(initial=0.0) == 0.0 and np.all(np.isinf(ex[e > 89.0]))
    sub = (e < -88.0) & (e > -103.0)
    assert (ex[sub] > 0).all()                                        # subnormal results are kept, not flushed
    v = I['log']
    lg, _ = orc.math_eval(3, v)
    okl = np.isfinite(v) & (v > 0)
    assert np.max(ulp_err(lg[okl], np.log(v[okl].astype(np.float64)))) <= 0.5 + 1e-6
    assert lg[v == 0][0] == -np.inf and lg[np.isinf(v)][0] == np.inf
    c = I['clamp']
    cl, _ = orc.math_eval(4, c)

    def ref_clamp(x):                                                  # constellation.cpp:263-270 (non-finite: 0, the x86 outcome)
        if not np.isfinite(x):
            return 0
        x = np.float32(x)
        while x < -127 or x > 127:
            x = np.float32(x * np.float32(0.5))
        return int(x)
    assert [int(t) for t in cl[-10:]] == [ref_clamp(t) for t in c[-10:]]
    assert np.array_equal(cl[:2000].astype(np.int32), np.array([ref_clamp(t) for t in c[:2000]], np.int32))


@pytest.mark.gpu
def test_device_evaluates_the_same_bits(engine):
    import torch
    rng = np.random.default_rng(2)
    I = inputs(rng, 2000000)

    def same(func, a, b=None, both=False):
        h0, h1 = orc.math_eval(func, a, b)
        d0, d1 = engine.math_eval(func, torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda() if b is not None else None)
        torch.cuda.synchronize()
        g0 = d0.cpu().numpy()
        nan = np.isnan(h0)
        assert np.array_equal(nan, np.isnan(g0))
        assert np.array_equal(h0.view(np.uint32)[~nan], g0.view(np.uint32)[~nan]), ('func', func, int((h0.view(np.uint32) != g0.view(np.uint32)).sum()))
        if both:
            assert np.array_equal(h1.view(np.uint32), d1.cpu().numpy().view(np.uint32)), ('func', func, 'second output')

    same(0, I['phase'], both=True)
    same(1, *I['xy'])
    same(2, I['exp'])
    same(3, I['log'])
    same(4, I['clamp'])
    # the LUT cell index: the kernels take it from ONE binary32 product-sum and fall back to the reference's double expression only within
    # 2.5e-4 of an integer -- equal to the double expression for every input, the cell boundaries and their float neighbours included
    edges = (np.arange(-300, 300, dtype=np.float64) * 1.5 / 256).astype(np.float32)
    near = np.concatenate([edges, np.nextafter(edges, np.float32(10)), np.nextafter(edges, np.float32(-10)),
                           (edges.astype(np.float64) + rng.uniform(-3e-6, 3e-6, edges.size)).astype(np.float32)])
    re = np.concatenate([rng.uniform(-0.8, 0.8, 2000000).astype(np.float32), near, rng.permutation(near),
                         np.float32([np.inf, -np.inf, np.nan, 1e30, -1e30, 0.0, -0.0, 0.75, -0.75])])
    im = np.concatenate([rng.uniform(-0.8, 0.8, 2000000).astype(np.float32), rng.permutation(near), near,
                         np.float32([0.1, 0.2, 0.3, np.nan, np.inf, -0.0, 0.0, -0.75, 0.75])])
    same(5, re, im)
    h0, _ = orc.math_eval(5, re, im)
    fin = np.isfinite(re) & np.isfinite(im)
    xi = np.clip(np.trunc(np.clip(re[fin].astype(np.float64) / 1.5 * 256 + 128, -1e9, 1e9)), 0, 255)
    yi = np.clip(np.trunc(np.clip(im[fin].astype(np.float64) / 1.5 * 256 + 128, -1e9, 1e9)), 0, 255)
    assert np.array_equal(h0[fin].astype(np.int64), (xi * 256 + yi).astype(np.int64))
