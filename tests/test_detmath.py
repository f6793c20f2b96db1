"""include/trace_detmath.h and include/trace_sampler.h: the deterministic elementary functions are (almost always) the
correctly rounded Float32 values — what Julia's Float32 sin/cos/tan/log give — and never more than 1 ulp away; the
library's host copy and the oracle's copy agree bit for bit; the sampler is uniform and reproducible."""
import ctypes as C

import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def orc(ob, fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    yy = np.ascontiguousarray(y, np.float32) if y is not None else x
    ob.lib().orc_detmath(fn, ob.fp(x), ob.fp(yy), x.size, ob.fp(out))
    return out


CASES = [
    (0, "sin", lambda x: np.sin(x), (-20.0, 20.0)),
    (1, "cos", lambda x: np.cos(x), (-20.0, 20.0)),
    (2, "tan", lambda x: np.tan(x), (-1.5, 1.5)),
    (4, "acos", lambda x: np.arccos(x), (-1.0, 1.0)),
    (5, "log", lambda x: np.log(x), (1e-6, 50.0)),
]


@pytest.mark.parametrize("fn,name,ref,dom", CASES)
def test_float32_functions_are_correctly_rounded(T, ob, fn, name, ref, dom):
    rng = np.random.default_rng(fn)
    x = rng.uniform(dom[0], dom[1], 400000).astype(np.float32)
    got = orc(ob, fn, x)
    want = ref(x.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, want)
    assert d.max() <= 1, f"{name}: {d.max()} ulp"
    assert (d != 0).mean() < 1e-5, f"{name}: {(d != 0).mean():.2e} of results are not the correctly rounded value"
    # the product library's host copy is the same function
    assert np.array_equal(T._ffi.detmath(fn, x).view(np.uint32), got.view(np.uint32))


def test_atan2(T, ob):
    rng = np.random.default_rng(7)
    x = rng.normal(size=300000).astype(np.float32)
    y = rng.normal(size=300000).astype(np.float32)
    got = orc(ob, 3, x, y)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, want)
    assert d.max() <= 1 and (d != 0).mean() < 1e-5
    assert np.array_equal(T._ffi.detmath(3, x, y).view(np.uint32), got.view(np.uint32))
    # special values used by compute_ϕ (sphere.jl:71-75)
    sp = orc(ob, 3, np.array([1, -1, 0, 0, -1], np.float32), np.array([0, 0, 1, -1, -0.0], np.float32))
    np.testing.assert_allclose(sp, [0, np.pi, np.pi / 2, -np.pi / 2, -np.pi], rtol=1e-7)


def test_float64_sin_cos(ob):
    x = np.random.default_rng(1).uniform(0, 2 * np.pi, 200000)
    for fn, ref in ((0, np.sin), (1, np.cos)):
        out = np.empty_like(x)
        ob.lib().orc_detmath_f64(fn, x.ctypes.data_as(C.POINTER(C.c_double)), x.size, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.max(np.abs(out - ref(x))) < 4e-16


def test_sampler_matches_numpy_model_and_is_uniform(T, ob):
    seed = 0x5EED0001
    px = np.array([-3, 0, 1, 17, 1025], np.int64)
    py = np.array([0, -1, 5, 900, 1025], np.int64)
    for s in (0, 1, 255):
        key = T.scenes.ts_stream_key(seed, px, py, s)
        for dim in (0, 4, 5, 12, 68):
            model = T.scenes.ts_uniform(key, dim)
            c = np.array([ob.lib().orc_sampler_u(seed, int(x), int(y), s, dim) for x, y in zip(px, py)], np.float32)
            assert np.array_equal(model.view(np.uint32), c.view(np.uint32))
    # uniformity / independence across dimensions and pixels
    X, Y = np.meshgrid(np.arange(256), np.arange(256))
    key = T.scenes.ts_stream_key(seed, X.ravel(), Y.ravel(), 3)
    u0, u1 = T.scenes.ts_uniform(key, 0), T.scenes.ts_uniform(key, 1)
    assert 0 <= u0.min() and u0.max() < 1
    assert abs(u0.mean() - 0.5) < 0.005 and abs(u0.var() - 1 / 12) < 0.002
    assert abs(np.corrcoef(u0, u1)[0, 1]) < 0.01
    assert abs(np.corrcoef(u0[:-1], u0[1:])[0, 1]) < 0.01


def test_sincos_equals_separate_calls(T):
    """tm_sincosf (one reduction, both kernels, quadrant by selection) returns tm_sinf / tm_cosf bit for bit, signed zeros,
    quadrant boundaries and non-finite arguments included."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-40, 40, 400000), rng.uniform(-1e3, 1e3, 100000), np.arange(-64, 65) * (np.pi / 4),
                        [0.0, -0.0, 1e-30, -1e-30, np.inf, -np.inf, np.nan, 3.1415927, 1.5707964, 6.2831855]]).astype(np.float32)
    s, c = T._ffi.detmath(0, x), T._ffi.detmath(1, x)
    s2, c2 = T._ffi.detmath(6, x), T._ffi.detmath(7, x)
    fin = np.isfinite(x)
    assert np.array_equal(s[fin].view(np.uint32), s2[fin].view(np.uint32))
    assert np.array_equal(c[fin].view(np.uint32), c2[fin].view(np.uint32))
    assert np.isnan(s2[~fin]).all() and np.isnan(c2[~fin]).all()


def test_python_sampler_protocol_matches_the_specification(T, ob):
    """The Python mirror's sampler walks UniformSampler's protocol (sampler/sampler.jl:129-151) over the seeded stream of
    include/trace_sampler.h: camera sample = dimensions 0-4, path vertex v starts at 5 + 8 v."""
    smp = T.SeededSampler(3, seed=0xABCDEF, sample_offset=7)
    L = ob.lib()
    smp.start_pixel((12, -3))
    n = 0
    while smp.has_next_sample():
        s = 7 + smp.current_sample - 1
        film, lens, time = smp.get_camera_sample(np.float32([12, -3]))
        want = [L.orc_sampler_u(0xABCDEF, 12, -3, s, d) for d in range(5)]
        assert film[0] == np.float32(12) + np.float32(want[0]) and film[1] == np.float32(-3) + np.float32(want[1])
        assert lens[0] == np.float32(want[2]) and lens[1] == np.float32(want[3]) and time == np.float32(want[4])
        for v in (0, 1, 4):
            smp.start_vertex(v)
            got = [smp.get_1d()] + list(smp.get_2d()) + list(smp.get_2d()) + list(smp.get_2d()) + [smp.get_1d()]
            assert got == [np.float32(L.orc_sampler_u(0xABCDEF, 12, -3, s, 5 + 8 * v + k)) for k in range(8)]
        smp.start_next_sample()
        n += 1
    assert n == 3
