"""An INDEPENDENT Float64 model of the reference's scattering functions, written from the Julia text — not from oracle/orc_scatter.h,
which it cross-checks (tests/test_oracle_bxdf_model.py; VERDICT r4 weak #6: product and oracle are two restatements by one author, and
the reference's own tests pin the BxDFs at normal incidence only).

Every function cites the Julia lines it follows (paths under /root/reference/src).  Plain Python floats (IEEE double); spectra are
3-tuples.  Decisions the reference takes on Float32 values (thresholds, `≈`) are taken here on doubles: a sample that sits within
`EDGE` of such a threshold is reported through `Model.edges` so that the comparison can set it aside instead of calling a branch
flip a disagreement.
"""
import math

PI = math.pi
RTOL32 = math.sqrt(2.0 ** -23)  # Julia's default rtol of `≈` for Float32: sqrt(eps(Float32))
EDGE = 2e-5

BSDF_NONE, BSDF_REFLECTION, BSDF_TRANSMISSION, BSDF_DIFFUSE, BSDF_GLOSSY, BSDF_SPECULAR, BSDF_ALL = 0, 1, 2, 4, 8, 16, 31  # reflection/bxdf.jl:1-7


class Edges:
    """Collects "this sample sits on a decision boundary" notes of one evaluation."""

    def __init__(self):
        self.notes = []

    def near(self, what, a, b, scale=None):
        s = max(abs(a), abs(b), 1e-30) if scale is None else scale
        if abs(a - b) <= EDGE * s:
            self.notes.append(what)


E = Edges()


# ---- vectors ---------------------------------------------------------------------------------------------------------------------
def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def mul(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def neg(a):
    return (-a[0], -a[1], -a[2])


def norm(a):
    return math.sqrt(dot(a, a))


def normalize(a):
    n = norm(a)
    return (a[0] / n, a[1] / n, a[2] / n)


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def smul(a, b):  # spectrum x spectrum
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def clamp(x, lo, hi):
    return hi if x > hi else (lo if x < lo else x)


def is_black(s):
    return s[0] == 0 and s[1] == 0 and s[2] == 0


ZERO = (0.0, 0.0, 0.0)


# ---- Trace.jl:48-67, 110-131, 147-163 ---------------------------------------------------------------------------------------------
def concentric_sample_disk(u):  # Trace.jl:48-61
    ox, oy = 2.0 * u[0] - 1.0, 2.0 * u[1] - 1.0
    if ox == 0.0 and oy == 0.0:  # `≈ 0` with the default tolerances holds for an exact zero only
        return (0.0, 0.0)
    E.near("disk |x| vs |y|", abs(ox), abs(oy))
    if abs(ox) > abs(oy):
        r, th = ox, (oy / ox) * PI / 4.0
    else:
        r, th = oy, PI / 2.0 - (ox / oy) * PI / 4.0
    return (r * math.cos(th), r * math.sin(th))


def cosine_sample_hemisphere(u):  # Trace.jl:63-67
    d = concentric_sample_disk(u)
    return (d[0], d[1], math.sqrt(max(0.0, 1.0 - d[0] * d[0] - d[1] * d[1])))


def cos_theta(w):  # Trace.jl:110
    return w[2]


def sin_theta2(w):
    return max(0.0, 1.0 - w[2] * w[2])


def sin_theta(w):
    return math.sqrt(sin_theta2(w))


def tan_theta(w):  # Trace.jl:113 (sin / cos: +-Inf when cos is zero)
    s, c = sin_theta(w), cos_theta(w)
    if c == 0.0:
        return math.copysign(math.inf, c) if s != 0 else math.nan
    return s / c


POLAR = 2e-2  # below this sin θ the Float32 value of sqrt(1 - cos²) is off by more than 1e-4 relative (1 - cos² is rounded to 6e-8 absolute), and so are cos ϕ, sin ϕ


def cos_phi(w):  # Trace.jl:115-118
    s = sin_theta(w)
    if 0.0 < s < POLAR:
        E.notes.append("phi of a near-polar direction")
    return 1.0 if s == 0.0 else clamp(w[0] / s, -1.0, 1.0)


def sin_phi(w):  # Trace.jl:119-122 (the degenerate value is 1 here too, as written)
    s = sin_theta(w)
    if 0.0 < s < POLAR:
        E.notes.append("phi of a near-polar direction")
    return 1.0 if s == 0.0 else clamp(w[1] / s, -1.0, 1.0)


def reflect(wo, n):  # Trace.jl:127
    return add(neg(wo), mul(n, 2.0 * dot(wo, n)))


def face_forward(n, v):  # Trace.jl:168
    return neg(n) if dot(n, v) < 0 else n


def spherical_direction(st, ct, phi):  # Trace.jl:147-149
    return (st * math.cos(phi), st * math.sin(phi), ct)


def same_hemisphere(w, wp):  # bxdf.jl:13-15
    return w[2] * wp[2] > 0


# ---- reflection/bxdf.jl:53-140 ------------------------------------------------------------------------------------------------------
def refract(wi, n, eta):  # bxdf.jl:53-63
    cos_i = dot(n, wi)
    sin2_i = max(0.0, 1.0 - cos_i * cos_i)
    sin2_t = eta * eta * sin2_i
    E.near("refract: total internal reflection", sin2_t, 1.0)
    if sin2_t >= 1:
        return False, ZERO
    cos_t = math.sqrt(1.0 - sin2_t)
    return True, add(mul(wi, -eta), mul(n, eta * cos_i - cos_t))


def fresnel_dielectric(cos_i, eta_i, eta_t):  # bxdf.jl:75-96
    cos_i = clamp(cos_i, -1.0, 1.0)
    E.near("fresnel: entering", cos_i, 0.0, 1.0)
    if cos_i <= 0.0:
        eta_i, eta_t = eta_t, eta_i
        cos_i = abs(cos_i)
    sin_i = math.sqrt(max(0.0, 1.0 - cos_i * cos_i))
    sin_t = sin_i * eta_i / eta_t
    E.near("fresnel: total internal reflection", sin_t, 1.0)
    if sin_t >= 1.0:
        return 1.0
    cos_t = math.sqrt(max(0.0, 1.0 - sin_t * sin_t))
    r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t)
    r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t)
    return 0.5 * (r_par * r_par + r_perp * r_perp)


class FresnelNoOp:  # bxdf.jl:140
    def __call__(self, c):
        return 1.0


class FresnelDielectric:  # bxdf.jl:133-139
    def __init__(self, eta_i, eta_t):
        self.eta_i, self.eta_t = eta_i, eta_t

    def __call__(self, c):
        return fresnel_dielectric(c, self.eta_i, self.eta_t)


class BxDF:
    type = 0

    def matches(self, flags):  # bxdf.jl:9-11
        return (self.type & flags) == self.type

    def pdf(self, wo, wi):  # bxdf.jl:23-25
        return abs(cos_theta(wi)) * (1.0 / PI) if same_hemisphere(wo, wi) else 0.0

    def sample_f(self, wo, u):  # bxdf.jl:34-42 -> (wi, pdf, f, sampled type or None)
        wi = cosine_sample_hemisphere(u)
        if wo[2] < 0:
            wi = (wi[0], wi[1], -wi[2])
        return wi, self.pdf(wo, wi), self.f(wo, wi), None


# ---- reflection/lambertian.jl ------------------------------------------------------------------------------------------------------
class LambertianReflection(BxDF):  # lambertian.jl:5-24
    def __init__(self, r):
        self.r, self.type = r, BSDF_DIFFUSE | BSDF_REFLECTION

    def f(self, wo, wi):
        return mul(self.r, 1.0 / PI)


class LambertianTransmission(BxDF):  # lambertian.jl:48-87
    def __init__(self, t):
        self.t, self.type = t, BSDF_DIFFUSE | BSDF_TRANSMISSION

    def f(self, wo, wi):
        return mul(self.t, 1.0 / PI)

    def sample_f(self, wo, u):  # :71-80
        wi = cosine_sample_hemisphere(u)
        if wo[2] > 0:
            wi = (wi[0], wi[1], -wi[2])
        return wi, self.pdf(wo, wi), self.f(wo, wi), None

    def pdf(self, wo, wi):  # :82-86
        return abs(cos_theta(wi)) * (1.0 / PI) if not same_hemisphere(wo, wi) else 0.0


# ---- reflection/specular.jl ----------------------------------------------------------------------------------------------------------
class SpecularReflection(BxDF):  # specular.jl:1-41
    def __init__(self, r, fresnel):
        self.r, self.fresnel, self.type = r, fresnel, BSDF_SPECULAR | BSDF_REFLECTION

    def f(self, wo, wi):
        return ZERO

    def sample_f(self, wo, u):  # :35-41
        wi = (-wo[0], -wo[1], wo[2])
        return wi, 1.0, mul(self.r, self.fresnel(cos_theta(wi)) / abs(cos_theta(wi))), None


class SpecularTransmission(BxDF):  # specular.jl:43-107
    def __init__(self, t, eta_a, eta_b):
        self.t, self.eta_a, self.eta_b = t, eta_a, eta_b
        self.fresnel = FresnelDielectric(eta_a, eta_b)
        self.type = BSDF_SPECULAR | BSDF_TRANSMISSION

    def f(self, wo, wi):
        return ZERO

    def sample_f(self, wo, u):  # :84-107
        entering = cos_theta(wo) > 0
        eta_i = self.eta_a if entering else self.eta_b
        eta_t = self.eta_b if entering else self.eta_a
        valid, wi = refract(wo, face_forward((0.0, 0.0, 1.0), wo), eta_i / eta_t)
        if not valid:
            return ZERO, 0.0, ZERO, None
        cos_wi = cos_theta(wi)
        k = 1.0 - self.fresnel(cos_wi)
        ft = mul(self.t, k)
        # `T isa Radiance && (ft *= ...)` (:104): T is a TYPE, never an instance of Radiance — the scaling does not happen
        return wi, 1.0, mul(ft, 1.0 / abs(cos_wi)), None


class FresnelSpecular(BxDF):  # specular.jl:110-173
    def __init__(self, r, t, eta_a, eta_b):
        self.r, self.t, self.eta_a, self.eta_b = r, t, eta_a, eta_b
        self.type = BSDF_SPECULAR | BSDF_TRANSMISSION | BSDF_REFLECTION

    def f(self, wo, wi):
        return ZERO

    def pdf(self, wo, wi):  # :138
        return 0.0

    def sample_f(self, wo, u):  # :144-173
        fd = fresnel_dielectric(cos_theta(wo), self.eta_a, self.eta_b)
        E.near("FresnelSpecular: u1 vs F", u[0], fd, 1.0)
        if u[0] < fd:
            wi = (-wo[0], -wo[1], wo[2])
            return wi, fd, mul(self.r, fd / abs(cos_theta(wi))), BSDF_SPECULAR | BSDF_REFLECTION
        if cos_theta(wo) > 0:
            eta_i, eta_t = self.eta_a, self.eta_b
        else:
            eta_i, eta_t = self.eta_b, self.eta_a
        ok, wi = refract(wo, face_forward((0.0, 0.0, 1.0), wo), eta_i / eta_t)
        if not ok:
            return wi, fd, ZERO, None
        pdf = 1.0 - fd
        ft = mul(self.t, pdf)  # (`T isa Radiance` again: no scaling)
        return wi, pdf, mul(ft, 1.0 / abs(cos_theta(wi))), BSDF_SPECULAR | BSDF_TRANSMISSION


# ---- reflection/microfacet.jl --------------------------------------------------------------------------------------------------------
class OrenNayar(BxDF):  # microfacet.jl:6-42
    def __init__(self, r, sigma_deg):
        s = math.radians(sigma_deg)
        s2 = s * s
        self.r = r
        self.a = 1.0 - (s2 / (2.0 * (s2 + 0.33)))
        self.b = 0.45 * s2 / (s2 + 0.09)
        self.type = BSDF_DIFFUSE | BSDF_REFLECTION

    def f(self, wo, wi):  # :22-42
        sin_i, sin_o = sin_theta(wi), sin_theta(wo)
        max_cos = 0.0
        E.near("OrenNayar: sin > 1e-4", sin_i, 1e-4)
        E.near("OrenNayar: sin > 1e-4", sin_o, 1e-4)
        if sin_i > 1e-4 and sin_o > 1e-4:
            d = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo)
            max_cos = max(0.0, d)
        # `if abs(cos_θ(wi) > abs(cos_θ(wo)))` (:34): abs of a Bool — the test is cos_θ(wi) > |cos_θ(wo)|, no abs on wi
        E.near("OrenNayar: cos_i vs |cos_o|", cos_theta(wi), abs(cos_theta(wo)), 1.0)
        if cos_theta(wi) > abs(cos_theta(wo)):
            sin_a, tan_b = sin_o, sin_i / abs(cos_theta(wi))
        else:
            sin_a, tan_b = sin_i, sin_o / abs(cos_theta(wo))
        return mul(self.r, (1.0 / PI) * (self.a + self.b * max_cos * sin_a * tan_b))


def roughness_to_alpha(rough):  # microfacet.jl:75-80
    rough = max(1e-3, rough)
    x = math.log(rough)
    return 1.62142 + 0.819955 * x + 0.1734 * x ** 2 + 0.0171201 * x ** 3 + 0.000640711 * x ** 4


class TrowbridgeReitz:  # microfacet.jl:53-201
    def __init__(self, ax, ay, sample_visible_area=True):
        self.ax, self.ay, self.sva = max(1e-3, ax), max(1e-3, ay), sample_visible_area

    def lam(self, w):  # :66-73
        t = abs(tan_theta(w))
        if math.isinf(t):
            return 0.0
        a = math.sqrt(cos_phi(w) ** 2 * self.ax ** 2 + sin_phi(w) ** 2 * self.ay ** 2)
        return (-1.0 + math.sqrt(1.0 + (a * t) ** 2)) / 2.0

    def G1(self, w):  # :82-84
        return 1.0 / (1.0 + self.lam(w))

    def G(self, wo, wi):  # :86-88
        return 1.0 / (1.0 + self.lam(wo) + self.lam(wi))

    def D(self, w):  # :94-101
        t2 = tan_theta(w) ** 2
        if math.isinf(t2):
            return 0.0
        c4 = cos_theta(w) ** 4
        e = (cos_phi(w) ** 2 / (self.ax ** 2) + sin_phi(w) ** 2 / (self.ay ** 2)) * t2
        return 1.0 / (PI * self.ax * self.ay * c4 * (1.0 + e) ** 2)

    def pdf(self, wo, wh):  # :103-106
        if not self.sva:
            return self.D(wh) * abs(cos_theta(wh))
        return self.D(wh) * self.G1(wo) * abs(dot(wo, wh)) / abs(cos_theta(wo))

    @staticmethod
    def _sample11(cos_t, u1, u2):  # :108-148
        E.near("TR sample: normal incidence", cos_t, 0.9999, 1.0)
        if cos_t > 0.9999:
            r = math.sqrt(u1 / (1.0 - u1))
            phi = 6.28318530718 * u2
            return r * math.cos(phi), r * math.sin(phi)
        sin_t = math.sqrt(max(0.0, 1.0 - cos_t ** 2))
        tan_t = sin_t / cos_t
        a = 1.0 / tan_t
        g1 = 2.0 / (1.0 + math.sqrt(1.0 + 1.0 / (a * a)))
        a = 2.0 * u1 / g1 - 1.0
        tmp = 1.0 / (a * a - 1.0)
        E.near("TR sample: tmp clamp", tmp, 1e10)
        if tmp > 1e10:
            tmp = 1e10
        b = tan_t
        b2 = b * b
        d = math.sqrt(max(0.0, b2 * tmp * tmp - (a * a - b2) * tmp))
        sx1, sx2 = b * tmp - d, b * tmp + d
        E.near("TR sample: slope choice", a, 0.0, 1.0)
        E.near("TR sample: slope choice", sx2, 1.0 / tan_t)
        slope_x = sx1 if (a < 0 or sx2 > 1.0 / tan_t) else sx2
        E.near("TR sample: u2 half", u2, 0.5, 1.0)
        if u2 > 0.5:
            s, u2 = 1.0, 2.0 * (u2 - 0.5)
        else:
            s, u2 = -1.0, 2.0 * (0.5 - u2)
        z = (u2 * (u2 * (u2 * 0.27385 - 0.73369) + 0.46341)) / (u2 * (u2 * (u2 * 0.093073 + 0.309420) - 1.0) + 0.597999)
        return slope_x, s * z * math.sqrt(1.0 + slope_x * slope_x)

    def _sample(self, wi, u1, u2):  # :150-166
        ws = normalize((wi[0] * self.ax, wi[1] * self.ay, wi[2]))
        sx, sy = self._sample11(cos_theta(ws), u1, u2)
        c, s = cos_phi(ws), sin_phi(ws)
        sx, sy = c * sx - s * sy, s * sx + c * sy
        sx *= self.ax
        sy *= self.ay
        return normalize((-sx, -sy, 1.0))

    def sample_wh(self, wo, u):  # :168-201 (sample_visible_area is always true in the materials: :62)
        assert self.sva
        flip = wo[2] < 0.0
        wh = self._sample(neg(wo) if flip else wo, u[0], u[1])
        return neg(wh) if flip else wh


class MicrofacetReflection(BxDF):  # microfacet.jl:204-258
    def __init__(self, r, dist, fresnel):
        self.r, self.d, self.fresnel, self.type = r, dist, fresnel, BSDF_REFLECTION | BSDF_GLOSSY

    def f(self, wo, wi):  # :222-236
        co, ci = abs(cos_theta(wo)), abs(cos_theta(wi))
        wh = add(wi, wo)
        if ci == 0 or co == 0:
            return ZERO
        if wh == ZERO:
            return ZERO
        wh = normalize(wh)
        fr = self.fresnel(dot(wi, face_forward(wh, (0.0, 0.0, 1.0))))
        return mul(self.r, self.d.D(wh) * self.d.G(wo, wi) * fr / (4.0 * ci * co))

    def sample_f(self, wo, u):  # :238-251
        if wo[2] == 0:
            return ZERO, 0.0, ZERO, None
        wh = self.d.sample_wh(wo, u)
        E.near("microfacet R: wo.wh sign", dot(wo, wh), 0.0, 1.0)
        if dot(wo, wh) < 0:
            return ZERO, 0.0, ZERO, None
        wi = reflect(wo, wh)
        E.near("microfacet R: hemisphere", wi[2], 0.0, 1.0)
        if not same_hemisphere(wo, wi):
            return ZERO, 0.0, ZERO, None
        pdf = self.pdf(wo, wh)  # (:249 hands compute_pdf the HALF vector where its signature says wi — kept as written)
        return wi, pdf, self.f(wo, wi), None

    def pdf(self, wo, wi):  # :253-258
        if not same_hemisphere(wo, wi):
            return 0.0
        wh = normalize(add(wo, wi))
        return self.d.pdf(wo, wh) / (4.0 * dot(wo, wh))


class MicrofacetTransmission(BxDF):  # microfacet.jl:261-337
    def __init__(self, t, dist, eta_a, eta_b):
        self.t, self.d, self.eta_a, self.eta_b = t, dist, eta_a, eta_b
        self.fresnel = FresnelDielectric(eta_a, eta_b)
        self.type = BSDF_TRANSMISSION | BSDF_GLOSSY

    def f(self, wo, wi):  # :281-307
        if same_hemisphere(wo, wi):
            return ZERO
        co, ci = cos_theta(wo), cos_theta(wi)
        if co == 0 or ci == 0:
            return ZERO
        eta = (self.eta_b / self.eta_a) if cos_theta(wo) > 0.0 else (self.eta_a / self.eta_b)
        wh = normalize(add(wo, mul(wi, eta)))
        if wh[2] < 0:
            wh = neg(wh)
        d_o, d_i = dot(wo, wh), dot(wi, wh)
        E.near("microfacet T: same side", d_o * d_i, 0.0, 1.0)
        if d_o * d_i > 0:
            return ZERO
        fr = self.fresnel(d_o)
        denom = d_o + eta * d_i
        factor = 1.0  # `T isa Radiance ? 1 / η : 1` (:298): never Radiance
        dd, dg = self.d.D(wh), self.d.G(wo, wi)
        k = abs(dd * dg * d_o * d_i * eta ** 2 * factor ** 2 / (ci * co * denom ** 2))
        return mul(self.t, (1.0 - fr) * k)

    def sample_f(self, wo, u):  # :309-322
        if wo[2] == 0:
            return ZERO, 0.0, ZERO, None
        wh = self.d.sample_wh(wo, u)
        E.near("microfacet T: wo.wh sign", dot(wo, wh), 0.0, 1.0)
        if dot(wo, wh) < 0:
            return ZERO, 0.0, ZERO, None
        eta = (self.eta_b / self.eta_a) if cos_theta(wo) > 0.0 else (self.eta_a / self.eta_b)
        ok, wi = refract(wo, wh, eta)
        if not ok:
            return ZERO, 0.0, ZERO, None
        return wi, self.pdf(wo, wi), self.f(wo, wi), None

    def pdf(self, wo, wi):  # :324-337
        if same_hemisphere(wo, wi):
            return 0.0
        eta = (self.eta_b / self.eta_a) if cos_theta(wo) > 0.0 else (self.eta_a / self.eta_b)
        wh = normalize(add(wo, mul(wi, eta)))
        d_o, d_i = dot(wo, wh), dot(wi, wh)
        E.near("microfacet T pdf: same side", d_o * d_i, 0.0, 1.0)
        if d_o * d_i > 0:
            return 0.0
        denom = d_o + eta * d_i
        return self.d.pdf(wo, wh) * abs(d_i * eta ** 2 / (denom ** 2))


# ---- materials/bsdf.jl ------------------------------------------------------------------------------------------------------------------
class BSDF:
    def __init__(self, ng, ns, dpdu, eta=1.0):  # bsdf.jl:41-51
        self.eta, self.ng, self.ns = eta, ng, ns
        self.ss = normalize(dpdu)
        self.ts = cross(ns, self.ss)
        self.bxdfs = []

    def to_local(self, v):  # :70-72
        return (dot(v, self.ss), dot(v, self.ts), dot(v, self.ns))

    def to_world(self, v):  # :74-76: Mat3f0(ss..., ts..., ns...) is column-major: columns ss, ts, ns
        return (self.ss[0] * v[0] + self.ts[0] * v[1] + self.ns[0] * v[2], self.ss[1] * v[0] + self.ts[1] * v[1] + self.ns[1] * v[2],
                self.ss[2] * v[0] + self.ts[2] * v[1] + self.ns[2] * v[2])

    def _select(self, b, flags, reflect):  # :91-96
        return b.matches(flags) and ((reflect and (b.type & BSDF_REFLECTION) != 0) or ((not reflect) and (b.type & BSDF_TRANSMISSION) != 0))

    def f(self, wo_w, wi_w, flags=BSDF_ALL):  # :79-100
        wo = self.to_local(wo_w)
        if wo[2] == 0.0:
            return ZERO
        wi = self.to_local(wi_w)
        g = dot(wi_w, self.ng) * dot(wo_w, self.ng)
        E.near("BSDF: reflect by ng", g, 0.0, 1.0)
        reflect = g > 0
        out = ZERO
        for b in self.bxdfs:
            if self._select(b, flags, reflect):
                out = add(out, b.f(wo, wi))
        return out

    def num_components(self, flags):  # :195-201
        return sum(1 for b in self.bxdfs if b.matches(flags))

    def sample_f(self, wo_w, u, flags):  # :107-175 -> (wi_world, f, pdf, sampled type)
        m = self.num_components(flags)
        if m == 0:
            return ZERO, ZERO, 0.0, BSDF_NONE
        x = u[0] * m
        E.near("BSDF: component choice", x, round(x), 1.0)
        comp = min(max(1, int(math.ceil(x))), m)
        count = comp
        comp -= 1
        chosen = None
        for b in self.bxdfs:
            if b.matches(flags):
                if count == 1:
                    chosen = b
                    break
                count -= 1
        u_re = (min(u[0] * m - comp, 1.0), u[1])
        wo = self.to_local(wo_w)
        if wo[2] == 0.0:
            return ZERO, ZERO, 0.0, BSDF_NONE
        sampled = chosen.type
        wi, pdf, f, st = chosen.sample_f(wo, u_re)
        if st is not None:
            sampled = st
        if pdf == 0.0:
            return ZERO, ZERO, 0.0, BSDF_NONE
        wi_w = self.to_world(wi)
        if not (chosen.type & BSDF_SPECULAR) and m > 1:  # :146-152
            for b in self.bxdfs:
                if b is not chosen and b.matches(flags):
                    pdf += b.pdf(wo, wi)
        if m > 1:
            pdf /= m
        if not (chosen.type & BSDF_SPECULAR):  # :155-167
            g = dot(wi_w, self.ng) * dot(wo_w, self.ng)
            E.near("BSDF: reflect by ng", g, 0.0, 1.0)
            reflect = g > 0
            f = ZERO
            for b in self.bxdfs:
                if self._select(b, flags, reflect):
                    f = add(f, b.f(wo, wi))
        return wi_w, f, pdf, sampled

    def pdf(self, wo_w, wi_w, flags):  # :177-193
        if not self.bxdfs:
            return 0.0
        wo = self.to_local(wo_w)
        if wo[2] == 0.0:
            return 0.0
        wi = self.to_local(wi_w)
        p, m = 0.0, 0
        for b in self.bxdfs:
            if b.matches(flags):
                m += 1
                p += b.pdf(wo, wi)
        return p / m if m > 0 else 0.0


# ---- materials/material.jl (constant textures) ---------------------------------------------------------------------------------------
def sclamp(s):  # spectrum.jl clamp(s): [0, Inf)
    return tuple(max(0.0, c) for c in s)


def matte(frame, kd, sigma, multi):  # material.jl:16-31
    b = BSDF(*frame)
    r = sclamp(kd)
    if is_black(r):
        return b
    s = clamp(sigma, 0.0, 90.0)
    b.bxdfs.append(LambertianReflection(r) if s == 0.0 else OrenNayar(r, s))
    return b


def mirror(frame, kr, multi):  # material.jl:39-46
    b = BSDF(*frame)
    r = sclamp(kr)
    if not is_black(r):
        b.bxdfs.append(SpecularReflection(r, FresnelNoOp()))
    return b


def glass(frame, kr, kt, ur, vr, index, remap, multi):  # material.jl:76-116
    b = BSDF(*frame, eta=index)
    r, t = sclamp(kr), sclamp(kt)
    if is_black(r) and is_black(t):
        return b
    is_specular = ur == 0 and vr == 0
    if is_specular and multi:
        b.bxdfs.append(FresnelSpecular(r, t, 1.0, index))
        return b
    if remap:
        ur, vr = roughness_to_alpha(ur), roughness_to_alpha(vr)
    dist = None if is_specular else TrowbridgeReitz(ur, vr)
    if not is_black(r):
        fr = FresnelDielectric(1.0, index)
        b.bxdfs.append(SpecularReflection(r, fr) if is_specular else MicrofacetReflection(r, dist, fr))
    if not is_black(t):
        b.bxdfs.append(SpecularTransmission(t, 1.0, index) if is_specular else MicrofacetTransmission(t, dist, 1.0, index))
    return b


def plastic(frame, kd, ks, rough, remap, multi):  # material.jl:135-151
    b = BSDF(*frame)
    d = sclamp(kd)
    if not is_black(d):
        b.bxdfs.append(LambertianReflection(d))
    s = sclamp(ks)
    if is_black(s):
        return b
    fr = FresnelDielectric(1.5, 1.0)
    if remap:
        rough = roughness_to_alpha(rough)
    b.bxdfs.append(MicrofacetReflection(s, TrowbridgeReitz(rough, rough), fr))
    return b
