"""Generates tests/golden/kat.json: known answers for the hit tests and shading helpers.

The reference ships no tests for the ray tracer, so these vectors are derived
here, independently of oracle/ and of the kernels, by evaluating the formulas of the cited
reference lines with numpy float32 scalars, one IEEE operation at a time (numpy never
contracts).  Run in the build container:  python tests/golden/make_golden.py
Floats are stored as u32 bit patterns so that comparisons are exact.
"""
import json
import os

import numpy as np

f32 = np.float32


def bits(x):
    return int(np.array(x, dtype=np.float32).view(np.uint32))


def vbits(v):
    return [bits(x) for x in v]


def F(x):
    return f32(x)


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def sub(a, b):
    return [a[0] - b[0], a[1] - b[1], a[2] - b[2]]


def at(o, d, t):
    return [o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]]


def vec(v):
    return [F(x) for x in v]


def sphere_hit(c, r, o, d, t_min, t_max):
    """objects/sphere.rs:45-82 + hittable.rs:17-41"""
    c, o, d, r, t_min, t_max = vec(c), vec(o), vec(d), F(r), F(t_min), F(t_max)
    oc = sub(o, c)
    a = dot(d, d)
    half_b = dot(oc, d)
    cc = dot(oc, oc) - r * r
    disc = half_b * half_b - a * cc
    if disc < 0:
        return None
    sq = np.sqrt(disc)
    root = (-half_b - sq) / a
    if root < t_min or t_max < root:
        root = (-half_b + sq) / a
        if root < t_min or t_max < root:
            return None
    p = at(o, d, root)
    n = [(p[i] - c[i]) / r for i in range(3)]
    front = dot(d, n) < 0
    nn = n if front else [-x for x in n]
    return {"t": bits(root), "normal": vbits(nn), "front_face": int(front)}


def rect_hit(kind, a0, a1, b0, b1, k, o, d, t_min, t_max):
    """rectangles.rs:22-45 (XY), 92-115 (XZ), 162-185 (YZ)"""
    A, B, AX = {"xy": (0, 1, 2), "xz": (0, 2, 1), "yz": (1, 2, 0)}[kind]
    a0, a1, b0, b1, k, t_min, t_max = F(a0), F(a1), F(b0), F(b1), F(k), F(t_min), F(t_max)
    o, d = vec(o), vec(d)
    with np.errstate(all="ignore"):
        t = (k - o[AX]) / d[AX]
        if t < t_min or t > t_max:
            return None
        a = o[A] + t * d[A]
        b = o[B] + t * d[B]
        if a < a0 or a > a1 or b < b0 or b > b1:
            return None
        n = [F(0), F(0), F(0)]
        n[AX] = F(1)
        front = dot(d, n) < 0
        nn = n if front else [-x for x in n]
        u = (a - a0) / (a1 - a0)
        v = (b - b0) / (b1 - b0)
    return {"t": bits(t), "normal": vbits(nn), "front_face": int(front), "uv": vbits([u, v])}


def aabb_hit(mn, mx, o, d, tmin, tmax):
    """aabb3.rs:19-45 with ffmin/ffmax of types.rs:171-185"""
    mn, mx, o, d, tmin, tmax = vec(mn), vec(mx), vec(o), vec(d), F(tmin), F(tmax)
    ffmin = lambda a, b: a if a < b else b
    ffmax = lambda a, b: a if a > b else b
    with np.errstate(all="ignore"):
        for a in range(3):
            inv = F(1) / d[a]
            t0 = ffmin((mn[a] - o[a]) * inv, (mx[a] - o[a]) * inv)
            t1 = ffmax((mn[a] - o[a]) * inv, (mx[a] - o[a]) * inv)
            tmin = ffmax(t0, tmin)
            tmax = ffmin(t1, tmax)
            if tmax <= tmin:
                return 0
    return 1


def normalize(v):
    lensq = dot(v, v)
    if lensq == 0:
        return [F(0)] * 3
    k = F(1) / np.sqrt(lensq)
    return [x * k for x in v]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def onb(n):
    """onb.rs:32-47"""
    w = normalize(vec(n))
    a = vec([0, 1, 0]) if abs(w[0]) > F(0.9) else vec([1, 0, 0])
    v = normalize(cross(w, a))
    u = cross(w, v)
    return vbits(u + v + w)


def schlick(cosine, ior):
    """types.rs:165-169; powi(5) = x * (x^2)^2"""
    cosine, ior = F(cosine), F(ior)
    r0 = (F(1) - ior) / (F(1) + ior)
    r0 = r0 * r0
    x = F(1) - cosine
    x2 = x * x
    x4 = x2 * x2
    return bits(r0 + (F(1) - r0) * (x * x4))


def refract(uv, n, eta):
    """math/src/vec3.rs:756-771"""
    uv, n, eta = vec(uv), vec(n), F(eta)
    cos_theta = dot([-x for x in uv], n)
    par = [eta * (uv[i] + cos_theta * n[i]) for i in range(3)]
    with np.errstate(all="ignore"):
        k = -np.sqrt(F(1) - dot(par, par))
    perp = [k * n[i] for i in range(3)]
    return vbits([par[i] + perp[i] for i in range(3)])


def smooth(t):
    return t * t * (F(3) - F(2) * t)


def perlin_noise(ranvec, px, py, pz, p):
    """perlin.rs:84-108 + 38-58: u,v,w smoothed in noise() AND again in perlin_interp()"""
    p = vec(p)
    fl = [np.floor(x) for x in p]
    u, v, w = [smooth(p[i] - fl[i]) for i in range(3)]
    i, j, k = [int(x) for x in fl]
    uu, vv, ww = smooth(u), smooth(v), smooth(w)
    acc = F(0)
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                idx = px[(i + di) & 255] ^ py[(j + dj) & 255] ^ pz[(k + dk) & 255]
                c = vec(ranvec[idx])
                wt = [u - F(di), v - F(dj), w - F(dk)]
                acc = acc + (F(di) * uu + (F(1) - F(di)) * (F(1) - uu)) * (F(dj) * vv + (F(1) - F(dj)) * (F(1) - vv)) * \
                    (F(dk) * ww + (F(1) - F(dk)) * (F(1) - ww)) * dot(c, wt)
    return acc


def turbulence(tab, p, depth=7):
    """perlin.rs:110-123"""
    p = vec(p)
    weight, acc = F(1), F(0)
    for _ in range(depth):
        acc = acc + weight * perlin_noise(*tab, p)
        weight = weight * F(0.5)
        p = [x * F(2) for x in p]
    return abs(acc)


def image_texel(w, h, u, v):
    """image_texture.rs:63-76 -> (i, j)"""
    u, v = F(u), F(v)
    clamp = lambda x: F(0) if x < 0 else (F(1) if x > 1 else x)
    u = clamp(u)
    v = F(1) - clamp(v)
    as_i32 = lambda x: 0 if np.isnan(x) else int(x)
    i, j = as_i32(u * F(w)), as_i32(v * F(h))
    return min(i, w - 1), min(j, h - 1)


def mat4_inverse_and_normal(m):
    """math/src/mat4.rs:536-617 (det with its two index typos, adjoint, invert) and transform.rs:171
    (normal2world = adjoint(world2object) transposed).  m: 4x4 list of f32, row-major."""
    def det(m):
        k1 = m[0][0] * m[1][1] - m[0][1] * m[1][0]; l1 = m[2][2] * m[3][3] - m[2][3] * m[3][2]
        k2 = m[0][0] * m[1][2] - m[0][2] * m[1][0]; l2 = m[2][1] * m[3][3] - m[1][2] * m[3][1]
        k3 = m[0][0] * m[1][3] - m[0][3] * m[1][0]; l3 = m[2][1] * m[3][2] - m[2][2] * m[3][1]
        k4 = m[0][1] * m[1][2] - m[0][2] * m[1][1]; l4 = m[2][0] * m[3][3] - m[3][2] * m[3][0]
        k5 = m[0][1] * m[1][3] - m[0][3] * m[1][1]; l5 = m[2][0] * m[3][2] - m[2][2] * m[3][0]
        k6 = m[0][2] * m[1][3] - m[0][3] * m[1][2]; l6 = m[2][0] * m[3][1] - m[2][1] * m[3][0]
        return k1 * l1 - k2 * l2 + k3 * l3 + k4 * l4 - k5 * l5 + k6 * l6

    def adjoint(m):
        m1 = m[2][2] * m[3][3] - m[2][3] * m[3][2]; m2 = m[2][1] * m[3][3] - m[2][3] * m[3][1]
        m3 = m[2][1] * m[3][2] - m[2][2] * m[3][1]; m4 = m[0][2] * m[1][3] - m[0][3] * m[1][2]
        m5 = m[0][1] * m[1][3] - m[0][3] * m[1][1]; m6 = m[0][1] * m[1][2] - m[0][2] * m[1][1]
        m7 = m[2][0] * m[3][3] - m[2][3] * m[3][0]; m8 = m[2][0] * m[3][2] - m[2][2] * m[3][0]
        m9 = m[1][2] * m[3][3] - m[1][3] * m[3][2]; m10 = m[1][0] * m[3][3] - m[1][3] * m[3][0]
        m11 = m[1][0] * m[3][2] - m[1][2] * m[3][0]; m12 = m[0][0] * m[1][3] - m[0][3] * m[1][0]
        m13 = m[0][0] * m[1][2] - m[0][2] * m[1][0]; m14 = m[2][0] * m[3][1] - m[2][1] * m[3][0]
        m15 = m[0][0] * m[1][1] - m[0][1] * m[1][0]; m16 = m[2][0] * m[3][1] - m[2][1] * m[3][0]
        return [[m[1][1] * m1 - m[1][2] * m2 + m[1][3] * m3, -m[0][1] * m1 + m[0][2] * m2 - m[0][3] * m3,
                 m[3][1] * m4 - m[3][2] * m5 + m[3][3] * m6, -m[2][1] * m4 + m[2][2] * m5 - m[2][3] * m6],
                [-m[1][0] * m1 + m[1][2] * m7 - m[1][3] * m8, m[0][0] * m1 - m[0][2] * m7 + m[0][3] * m8,
                 -m[0][0] * m9 + m[0][2] * m10 - m[0][3] * m11, m[2][0] * m4 - m[2][2] * m12 + m[2][3] * m13],
                [m[1][0] * m2 - m[1][1] * m7 + m[1][3] * m16, -m[0][0] * m2 + m[0][1] * m7 - m[0][3] * m16,
                 m[3][0] * m5 - m[3][1] * m12 + m[3][3] * m15, -m[2][0] * m5 + m[2][1] * m12 - m[2][3] * m15],
                [-m[1][0] * m3 + m[1][1] * m8 - m[1][2] * m14, m[0][0] * m3 - m[0][1] * m8 + m[0][2] * m14,
                 -m[3][0] * m6 + m[3][1] * m13 - m[3][2] * m15, m[2][0] * m6 - m[2][1] * m13 + m[2][2] * m15]]
    d = det(m)
    inv = F(1) / d
    w2o = [[e * inv for e in row] for row in adjoint(m)]
    adj = adjoint(w2o)
    n2w = [[adj[c][r] for c in range(4)] for r in range(4)]
    return w2o, n2w


def mat_vec(m, v, w):
    """math/src/mat4.rs:434-441 followed by .xyz()"""
    w = F(w)
    return [m[r][0] * v[0] + m[r][1] * v[1] + m[r][2] * v[2] + m[r][3] * w for r in range(3)]


def poly_quadratic(a, b, c):
    """math/src/polynomial.rs:11-39 -> (count, [r0, r1]) with the caller's zero-initialised roots"""
    zero = F(0)
    if a == 0:
        if b == 0:
            return 0, [zero, zero]
        return 1, [-c / b, zero]
    delta = b * b - F(4) * a * c
    if delta == 0:
        r = -b / F(2) * a
        return 2, [r, r]
    if not delta < 0:
        sgn = np.copysign(F(1), b)
        q = -(b + sgn * np.sqrt(delta)) / F(2)
        r0, r1 = q / a, c / q
        if r0 > r1:
            r0, r1 = r1, r0
        return 2, [r0, r1]
    return 0, [zero, zero]


def quadric_hit(kind, prm, m, o, d, t_min, t_max):
    """cylinder.rs:52-128, cone.rs:35-121, paraboloid.rs:48-144 under an optional Transform (transform.rs:184-204).
    phi only decides the clipping here (its bits depend on the atan2 in use), so cases keep phi_max = 2 pi or stay clear of it."""
    o, d, t_min, t_max = vec(o), vec(d), F(t_min), F(t_max)
    prm = [F(x) for x in prm]
    o_w, d_w = o, d
    if m is not None:
        m = [[F(x) for x in row] for row in m]
        w2o, n2w = mat4_inverse_and_normal(m)
        o, d = mat_vec(w2o, o, 1), mat_vec(w2o, d, 0)
    two, two_pi = F(2), F(np.pi) * F(2)
    with np.errstate(all="ignore"):
        if kind == "cylinder":
            radius, zmin, zmax, phi_max = prm
            a = d[0] * d[0] + d[1] * d[1]
            b = two * (d[0] * o[0] + d[1] * o[1])
            c = o[0] * o[0] + o[1] * o[1] - radius * radius
        elif kind == "cone":
            radius, height, phi_max = prm
            zmin, zmax = F(0), height
            k = radius / height
            k = k * k
            a = d[0] * d[0] + d[1] * d[1] - k * d[2] * d[2]
            b = two * (d[0] * o[0] + d[1] * o[1] - k * d[2] * (o[2] - height))
            c = o[0] * o[0] + o[1] * o[1] - k * (o[2] - height) * (o[2] - height)
        else:
            radius, zmin, zmax, phi_max = prm
            k = zmax / (radius * radius)
            a = k * (d[0] * d[0] + d[1] * d[1])
            b = two * k * (d[0] * o[0] + d[1] * o[1]) - d[2]
            c = k * (o[0] * o[0] + o[1] * o[1]) - o[2]
        n, (t0, t1) = poly_quadratic(a, b, c)
        if n == 0:
            return None
        if kind == "cylinder":
            if t0 > t_max or t1 < t_min:
                return None
            thit = t1 if t0 < t_min else t0
        else:
            if t0 > t_max or t1 <= 0:
                return None
            thit = t1 if t0 <= 0 else t0
        if thit > t_max:
            return None

        def phi_of(p):
            phi = F(np.arctan2(p[1], p[0]))
            return phi + two_pi if phi < 0 else phi
        p = at(o, d, thit)
        phi = phi_of(p)
        if p[2] < zmin or p[2] > zmax or phi > phi_max:
            if thit == t1:
                return None
            thit = t1
            if t1 > t_max:
                return None
            p = at(o, d, thit)
            phi = phi_of(p)
            if p[2] < zmin or p[2] > zmax or phi > phi_max:
                return None
        zero = F(0)
        if kind == "cylinder":
            v = (p[2] - zmin) / (zmax - zmin)
            dpdu, dpdv = [-phi_max * p[1], phi_max * p[0], zero], [zero, zero, zmax - zmin]
            n_out = normalize(cross(dpdv, dpdu))
        elif kind == "cone":
            v = p[2] / height
            dpdu = [-phi_max * p[1], phi_max * p[1], zero]
            dpdv = [-p[0] / (F(1) - v), -p[1] / (F(1) - v), height]
            n_out = normalize(cross(dpdu, dpdv))
        else:
            v = (p[2] - zmin) / (zmax - zmin)
            dpdu = [-phi_max * p[1], phi_max * p[0], zero]
            s = zmax - zmin
            dpdv = [s * (p[0] / (two * p[2])), s * (p[1] / (two * p[2])), s * F(1)]
            n_out = normalize(cross(dpdu, dpdv))
        front = dot(d, n_out) < 0
        nn = n_out if front else [-x for x in n_out]
        if m is not None:
            p = mat_vec(m, p, 1)
            nn = normalize(mat_vec(n2w, nn, 0))
    return {"t": bits(thit), "p": vbits(p), "normal": vbits(nn), "front_face": int(front), "v": bits(v)}


def main():
    out = {}
    sph = []
    for case in [
        ([0, 0, 0], 1, [0, 0, -3], [0, 0, 1], 0.001, np.inf),        # t = 2 exactly
        ([0, 0, 0], 1, [0, 0, 0], [0, 0, 2], 0.001, np.inf),         # inside: second root, back face
        ([0, 0, 0], 1, [0, 2, -3], [0, 0, 1], 0.001, np.inf),        # miss
        ([0, 0, 0], 1, [0, 0, -3], [0, 0, 1], 0.001, 1.5),           # both roots outside the range
        ([0, 1, -3], 1, [0, 0, -3], [0.1, 0.9, 0.3], 0.001, np.inf),
        ([190, 90, 190], 90, [278, 278, -800], [-0.11, -0.2, 1.0], 0.001, np.inf),   # Cornell glass sphere
        ([0, -1000, 0], 1000, [13, 2, 3], [-13, -2.2, -3.1], 0.001, np.inf),         # Book-1 ground
        ([0, 0, 0], 1, [0, 1, -3], [0, 0, 1], 0.001, np.inf),        # grazing: disc == 0
    ]:
        sph.append({"center": case[0], "radius": case[1], "o": case[2], "d": case[3], "t_min": case[4],
                    "t_max": "inf" if case[5] == np.inf else case[5], "hit": sphere_hit(*case)})
    out["sphere"] = sph
    rects = []
    for case in [
        ("xz", 213, 343, 227, 332, 554, [278, 278, 278], [0, 1, 0], 0.001, np.inf),   # Cornell light from below: front (n=+y, d=+y -> back)
        ("xz", 213, 343, 227, 332, 554, [278, 600, 278], [0.1, -1, 0.05], 0.001, np.inf),
        ("xy", 0, 555, 0, 555, 555, [278, 278, -800], [0.2, -0.1, 1], 0.001, np.inf),
        ("yz", 0, 555, 0, 555, 0, [278, 278, -800], [-1, 0.1, 2.5], 0.001, np.inf),
        ("xy", 0, 1, 0, 1, 0, [0.5, 0.5, -1], [0.5, 0.5, 1], 0.001, np.inf),          # exactly on the corner x=1,y=1: inclusive
        ("xy", 0, 1, 0, 1, 0, [0.5, 0.5, -1], [0.6, 0.5, 1], 0.001, np.inf),          # just outside
        ("xy", 0, 1, 0, 1, 0, [0.5, 0.5, 0], [1, 0, 0], 0.001, np.inf),               # ray in the plane: 0/0 -> NaN t is ACCEPTED
        ("xy", 0, 1, 0, 1, 0, [0.5, 0.5, 1], [1, 0, 0], 0.001, np.inf),               # parallel, off the plane: -inf -> rejected
    ]:
        rects.append({"kind": case[0], "a0": case[1], "a1": case[2], "b0": case[3], "b1": case[4], "k": case[5], "o": case[6],
                      "d": case[7], "t_min": case[8], "t_max": "inf", "hit": rect_hit(*case)})
    out["rect"] = rects
    boxes = []
    for case in [
        ([0, 0, 0], [1, 1, 1], [0.5, 0.5, -1], [0, 0, 1], 0.001, np.inf),
        ([0, 0, 0], [1, 1, 1], [0.5, 0.5, -1], [0.1, 0.1, 1], 0.001, np.inf),
        ([0, 0, 0], [1, 1, 1], [2, 0.5, -1], [0.1, 0.1, 1], 0.001, np.inf),
        ([0, 0, 0], [1, 1, 1], [0.5, 0.5, -1], [0.1, 0.1, 1], 0.001, 0.5),     # box beyond t_max
        ([0, 0, 0], [1, 1, 0], [0.5, 0.5, -1], [0.1, 0.1, 1], 0.001, np.inf),  # zero thickness: tmax <= tmin -> miss
        ([0, 0, 0], [1, 1, 1], [0, 0.5, -1], [0, 0.1, 1], 0.001, np.inf),      # origin on the min plane, d.x = 0: 0*inf = NaN
        ([0, 0, 0], [1, 1, 1], [1.5, 0.5, -1], [0, 0.1, 1], 0.001, np.inf),    # d.x = 0 outside the slab
        ([0, 0, 0], [1, 1, 1], [0.5, 0.5, 2], [0.1, 0.1, 1], 0.001, np.inf),   # box behind the ray
    ]:
        boxes.append({"min": case[0], "max": case[1], "o": case[2], "d": case[3], "t_min": case[4],
                      "t_max": "inf" if case[5] == np.inf else case[5], "hit": aabb_hit(*case)})
    out["aabb"] = boxes
    out["onb"] = [{"n": n, "uvw": onb(n)} for n in ([0, 1, 0], [1, 0, 0], [0.95, 0.1, 0.2], [0.3, -0.4, 0.8], [0, 0, -2])]
    out["schlick"] = [{"cosine": c, "ior": i, "value": schlick(c, i)} for c, i in ((1.0, 1.5), (0.0, 1.5), (0.3, 0.6666667), (0.7, 1.5))]
    out["refract"] = [{"uv": u, "n": n, "eta": e, "value": refract(u, n, e)} for u, n, e in (
        ([0, -1, 0], [0, 1, 0], 0.6666667), ([0.6, -0.8, 0], [0, 1, 0], 0.6666667), ([0.6, -0.8, 0], [0, 1, 0], 1.5))]
    # Perlin with a synthetic table: permutations i -> (i*k + c) mod 256, gradients on a fixed pattern
    px = [(i * 7 + 3) % 256 for i in range(256)]
    py = [(i * 11 + 5) % 256 for i in range(256)]
    pz = [(i * 13 + 1) % 256 for i in range(256)]
    ran = []
    for i in range(256):
        v = vec([np.sin(F(i) * F(0.37)), np.cos(F(i) * F(0.91)), np.sin(F(i) * F(1.3) + F(0.5))])
        ran.append([float(x) for x in normalize(v)])
    tab = (ran, px, py, pz)
    pts = [[0.5, 0.5, 0.5], [1.25, -2.75, 3.125], [22.0, 28.0, 30.0], [-0.001, 100.7, 7.3], [220.0, 280.0, 380.0]]
    out["perlin"] = {"ranvec": [vbits(r) for r in ran], "perm_x": px, "perm_y": py, "perm_z": pz,
                     "noise": [{"p": p, "value": bits(perlin_noise(*tab, p))} for p in pts],
                     "turbulence": [{"p": p, "value": bits(turbulence(tab, p))} for p in pts]}
    out["image"] = [{"w": 4, "h": 2, "u": u, "v": v, "ij": list(image_texel(4, 2, u, v))} for u, v in (
        (0.0, 0.0), (1.0, 1.0), (0.999, 0.001), (0.26, 0.74), (-3.0, 7.0), (0.5, 0.5), (float("nan"), 0.25))]
    # the reference's own adjacent known answer: math/src/mat4.rs:641-655
    out["mat4_mul"] = {"a": list(range(1, 17)), "b": list(range(17, 33)),
                       "result": [250, 260, 270, 280, 618, 644, 670, 696, 986, 1028, 1070, 1112, 1354, 1412, 1470, 1528]}
    # quadrics, alone and under general Transforms (rotation about a tilted axis, non-uniform scale, translation)
    def trs(t, angle_deg, axis, scale):
        ax = np.asarray(axis, np.float64) / np.linalg.norm(axis)
        c, si = np.cos(np.radians(angle_deg)), np.sin(np.radians(angle_deg))
        x, y, z = ax
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * si, x * z * (1 - c) + y * si, 0],
                      [y * x * (1 - c) + z * si, c + y * y * (1 - c), y * z * (1 - c) - x * si, 0],
                      [z * x * (1 - c) - y * si, z * y * (1 - c) + x * si, c + z * z * (1 - c), 0], [0, 0, 0, 1]])
        tm = np.eye(4); tm[:3, 3] = t
        return [[float(F(e)) for e in row] for row in (tm @ r @ np.diag(list(scale) + [1.0]))]
    two_pi = float(F(np.pi) * F(2))
    shapes = [("cylinder", [1.0, -0.5, 0.5, two_pi]), ("cone", [1.0, 1.0, two_pi]), ("paraboloid", [0.5, -0.5, 1.0, two_pi]),
              ("cylinder", [0.7, -1.0, 2.0, two_pi]), ("cone", [0.5, 2.0, two_pi]), ("paraboloid", [1.0, 0.0, 1.0, two_pi])]
    mats = [None, trs((3, 1, -2), 37, (1, 2, 3), (1.5, 0.5, 2.0)), trs((-1, 0.5, 4), -110, (0, 1, 0.3), (2, 2, 2)), trs((0, 0, 0), 63, (1, 0, 0), (1, 1, 1))]
    rng = np.random.default_rng(12345)
    quads = []
    for si, (kind, prm) in enumerate(shapes):
        for mi, m in enumerate(mats):
            kept = 0
            while kept < 3:
                # aim at a point near the shape from a point a few units away (world space)
                centre = np.array([0.0, 0.0, 0.3]) if m is None else (np.asarray(m, np.float64) @ np.array([0, 0, 0.3, 1.0]))[:3]
                o = centre + rng.normal(size=3) * 4.0
                target = centre + rng.normal(size=3) * (0.5 if m is None else 1.0)
                dd = target - o
                o_, d_ = [float(F(x)) for x in o], [float(F(x)) for x in dd]
                t_min, t_max = (0.001, float("inf")) if kept != 2 else (0.001, 1.5)
                h = quadric_hit(kind, prm, m, o_, d_, t_min, t_max)
                if h is None and kept == 0:
                    continue                                   # the first case of every pair is a hit
                quads.append({"kind": kind, "params": prm, "matrix": m, "o": o_, "d": d_, "t_min": t_min,
                              "t_max": "inf" if t_max == float("inf") else t_max, "hit": h})
                kept += 1
    out["quadric"] = quads
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
