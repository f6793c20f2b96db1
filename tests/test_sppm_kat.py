"""Known answers for the callees only SPPM uses (oracle/orc_sppm.h), derived by hand from the reference's formulas:
radical_inverse / Distribution1D / sample_discrete (sampler/sampling.jl:3-60), to_grid / hash (integrators/sppm.jl:479-501),
sample_le (lights/point.jl:60-69).  The reference has no test for any of them; these pin the restatement to the text.
"""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32


def odd_primes(n):
    out, c = [], 3
    while len(out) < n:
        if all(c % d for d in range(3, int(c ** 0.5) + 1, 2)):
            out.append(c)
        c += 2
    return out


def radical_inverse_model(dim, a):
    """sampling.jl:43-60 transcribed with numpy scalars: Float32 inverse-base powers, Float64 digit division."""
    if dim == 0:
        rev = int(f"{a:064b}"[::-1], 2)
        return f32(np.float64(rev) * np.float64(5.4210108624275222e-20))
    base = odd_primes(dim)[-1]  # PRIMES omits 2: PRIMES[1] = 3
    inv_base = f32(1.0) / f32(base)
    rev, inv_base_n = 0, f32(1.0)
    while a > 0:
        nxt = int(np.floor(np.float64(a) / np.float64(base)))
        rev = rev * base + (a - nxt * base)
        inv_base_n = f32(inv_base_n * inv_base)
        a = nxt
    return min(f32(f32(rev) * inv_base_n), f32(1.0))


def test_radical_inverse_closed_forms(ob):
    ri = ob.lib().orc_radical_inverse
    assert ri(0, 1) == 0.5 and ri(0, 2) == 0.25 and ri(0, 3) == 0.75 and ri(0, 0) == 0.0
    assert f32(ri(1, 1)) == f32(1.0) / f32(3.0)                # base 3: "1" -> 0.1
    assert f32(ri(1, 3)) == f32(f32(1.0) / f32(3.0)) * f32(f32(1.0) / f32(3.0))  # "10" -> 0.01 (base 3)
    assert abs(ri(2, 7) - 11.0 / 25.0) < 1e-7                  # base 5: "12" -> 0.21
    assert abs(ri(3, 50) - 50.0 / 343.0) < 1e-7                # base 7: "101" -> 0.101 = 1/7 + 1/343


def test_radical_inverse_matches_transcription(ob):
    ri = ob.lib().orc_radical_inverse
    for dim in list(range(0, 12)) + [29, 53, 100, 191]:
        for a in list(range(0, 300)) + [1023, 1046529, 104652900 - 1, 2 ** 31 + 12345, 2 ** 40 + 7]:
            got = f32(ri(dim, a))
            want = radical_inverse_model(dim, a)
            assert got.view(np.uint32) == f32(want).view(np.uint32), (dim, a, got, want)


def test_grid_hash(ob):
    h = ob.lib().orc_grid_hash
    for x, y, z, n in [(0, 0, 0, 1000), (1, 2, 3, 1000), (21, 20, 20, 1048576), (2 ** 40, 5, 9, 1046529)]:
        want = (((x * 73856093) & (2 ** 64 - 1)) ^ ((y * 19349663) & (2 ** 64 - 1)) ^ ((z * 83492791) & (2 ** 64 - 1))) % n + 1
        assert h(x, y, z, n) == want


def test_distribution1d_and_sample_discrete(ob):
    L = ob.lib()
    func = np.float32([1.0, 3.0])
    cdf = np.empty(3, np.float32)
    func_int = L.orc_distribution1d(ob.fp(func), 2, ob.fp(cdf))
    assert func_int == 2.0 and list(cdf) == [0.0, 0.25, 1.0]
    out = np.empty(3, np.float32)
    L.orc_sample_discrete(ob.fp(func), 2, 0.2, ob.fp(out))
    assert out[0] == 1 and out[1] == 0.25 and out[2] == f32(0.2) / f32(0.25)
    L.orc_sample_discrete(ob.fp(func), 2, 0.25, ob.fp(out))  # findlast(cdf ≤ u): the boundary belongs to the upper interval
    assert out[0] == 2 and out[1] == 0.75 and out[2] == 0.0
    L.orc_sample_discrete(ob.fp(func), 2, 1.0, ob.fp(out))   # offset clamped to n
    assert out[0] == 2
    zero = np.float32([0.0, 0.0, 0.0])
    cdf = np.empty(4, np.float32)
    assert L.orc_distribution1d(ob.fp(zero), 3, ob.fp(cdf)) == 0.0
    assert list(cdf) == [0.0, f32(2 / 3), f32(3 / 3), f32(4 / 3)]  # `cdf[i] = i / n` with the 1-based i in 2:n+1 (sampling.jl:19-22)


def test_to_grid(ob):
    L = ob.lib()
    bounds = np.float32([0, 0, 0, 2, 4, 8])
    res = np.int64([2, 4, 8])
    out = np.empty(4, np.int64)
    i64 = C.POINTER(C.c_int64)

    def grid(p):
        L.orc_to_grid(ob.fp(np.float32(p)), ob.fp(bounds), res.ctypes.data_as(i64), out.ctypes.data_as(i64))
        return list(out)

    assert grid([1, 1, 1]) == [1, 1, 1, 1]
    assert grid([0, 0, 0]) == [1, 0, 0, 0]
    assert grid([2, 4, 8]) == [0, 1, 3, 7]        # offset 1 -> floor(res) = res: out of bounds, clamped to res - 1
    assert grid([-1, 1, 1]) == [0, 0, 1, 1]       # negative cell: out of bounds, clamped to 0
    assert grid([1.999, 3.999, 7.999]) == [1, 1, 3, 7]
    flat = np.float32([0, 0, 0, 2, 0, 8])         # degenerate axis: offset divides by 1 (bounds.jl:134-143)
    L.orc_to_grid(ob.fp(np.float32([1, 0, 4])), ob.fp(flat), res.ctypes.data_as(i64), out.ctypes.data_as(i64))
    assert list(out) == [1, 1, 0, 4]


def test_sample_le_point_light(T, ob):
    scene = T.scenes.cornell_scene()
    osc = ob.OracleScene.from_scene(scene)
    out = np.empty(11, np.float32)
    assert ob.lib().orc_sample_le(osc.h, 0, ob.fp(np.float32([0.5, 0.25])), ob.fp(out)) == 0
    assert list(out[:3]) == [2.5, 2.5, 2.5] and list(out[3:6]) == [0.5, f32(0.9), -2.5]
    # uniform_sample_sphere(0.5, 0.25): z = 0, r = 1, ϕ = π/2 (Trace.jl:69-74)
    assert abs(out[6]) < 1e-6 and out[7] == 1.0 and out[8] == 0.0
    assert out[9] == 1.0 and out[10] == f32(1.0) / (f32(4.0) * f32(np.pi))


def test_sppm_first_iteration_update_formula(T, ob):
    """_update_pixels! after one iteration (sppm.jl:438-459) from its formulas: N = Float32(2/3) * M (Float32 product, widened),
    radius = Float32(r0 * sqrt(N / M)) evaluated in Float64, τ = (0 + ϕ) * (radius_new / r0)^2 rounded once."""
    scene = T.scenes.cornell_scene()
    cam = T.scenes.cornell_camera(24)
    osc = ob.OracleScene.from_scene(scene)
    r0 = np.float32(0.09)
    r = osc.sppm(cam, r0, 4, 1, 4000, seed=2)
    M, N, rad = r["M"], r["N"], r["radius"]
    hit = M > 0
    assert hit.sum() > 50
    n_expect = (np.float32(2.0) / np.float32(3.0) * M[hit].astype(np.float32)).astype(np.float64)
    assert np.array_equal(N[hit], n_expect)
    rad_new64 = np.float64(r0) * np.sqrt(n_expect / M[hit].astype(np.float64))
    assert np.array_equal(rad[hit], rad_new64.astype(np.float32))
    assert np.all(rad[~hit] == r0) and np.all(N[~hit] == 0)
    ratio2 = (rad_new64 / np.float64(r0)) ** 2
    tau_expect = (r["phi"][hit].astype(np.float32).astype(np.float64) * ratio2[:, None]).astype(np.float32)
    assert np.array_equal(r["tau"][hit], tau_expect)
