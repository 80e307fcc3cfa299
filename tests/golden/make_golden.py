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


# ---------------------------------------------------------------------------------------------------------------------
# Round 2: the parts of the path that round 1 left resting on one reading of the Rust (VERDICT r1 "weak" 1): triangle test
# and cull, MovingSphere, Translate / RotateY records, and one whole bounce of ray_color for metal, dielectric, Lambertian
# (mixture pdf, its clamp, rectangle and sphere pdf_value) and ConstantMedium::hit with given random numbers.

def scale(k, v):
    return [k * x for x in v]


def add(a, b):
    return [a[0] + b[0], a[1] + b[1], a[2] + b[2]]


def neg(v):
    return [-x for x in v]


def length(v):
    return np.sqrt(dot(v, v))


def hit_record_new(d, outward):
    """hittable.rs:17-41: (normal, front_face)"""
    front = bool(dot(d, outward) < 0)
    return (outward if front else neg(outward)), front


def tri_hit(verts, o, d, t_min, t_max):
    """triangle_mesh.rs:185-243 (ray_triangle_intersect_test); verts = 3 x (pos, normal, uv)"""
    (p1, n1, uv1), (p2, n2, uv2), (p3, n3, uv3) = [(vec(v[0:3]), vec(v[3:6]), vec(v[6:8])) for v in verts]
    o, d, t_min, t_max = vec(o), vec(d), F(t_min), F(t_max)
    e1, e2 = sub(p2, p1), sub(p3, p1)
    s1 = cross(d, e2)
    div = dot(s1, e1)
    if div == 0:
        return None
    inv_div = F(1) / div
    dd = sub(o, p1)
    b1 = dot(dd, s1) * inv_div
    if b1 < 0 or b1 > 1:
        return None
    s2 = cross(dd, e1)
    b2 = dot(d, s2) * inv_div
    if b2 < 0 or (b1 + b2) > 1:
        return None
    t = dot(e2, s2) * inv_div
    if t < t_min or t > t_max:
        return None
    b0 = F(1) - b1 - b2
    n = normalize(add(add(scale(b0, n1), scale(b1, n2)), scale(b2, n3)))
    if dot(d, n) > 0:
        return None
    uv = [b0 * uv1[i] + b1 * uv2[i] + b2 * uv3[i] for i in range(2)]
    nn, front = hit_record_new(d, n)
    return {"t": bits(t), "normal": vbits(nn), "front_face": int(front), "uv": vbits(uv)}


def mesh_hit(verts, o, d, t_min, t_max):
    """TriangleMesh::hit (triangle_mesh.rs:245-268) for a one-triangle, one-node mesh: both boxes, then the triangle"""
    pts = np.array([v[0:3] for v in verts], np.float32)
    mn, mx = pts.min(axis=0), pts.max(axis=0)
    if not aabb_hit(mn, mx, o, d, t_min, t_max):
        return None
    return tri_hit(verts, o, d, t_min, t_max)


def moving_center(c0, c1, t0, t1, time):
    """sphere.rs:144-147"""
    k = (F(time) - F(t0)) / (F(t1) - F(t0))
    return add(vec(c0), scale(k, sub(vec(c1), vec(c0))))


def moving_sphere_hit(c0, c1, t0, t1, r, o, d, time, t_min, t_max):
    """sphere.rs:151-201"""
    return sphere_hit([float(x) for x in moving_center(c0, c1, t0, t1, time)], r, o, d, t_min, t_max)


def wrapped_hit(inner, sin_t, cos_t, offset, o, d, t_min, t_max):
    """Translate(RotateY(obj)) (transform.rs:28-45, 116-144).  inner(o, d, t_min, t_max) -> (t, p, outward_normal_as_returned,
    front_face) of the wrapped object in ITS frame (its own HitRecord::new already applied), or None."""
    o, d, off, s, c = vec(o), vec(d), vec(offset), F(sin_t), F(cos_t)
    to = sub(o, off)                                                   # Translate: the ray moves back
    ro = [c * to[0] - s * to[2], to[1], s * to[0] + c * to[2]]         # RotateY :117-124
    rd = [c * d[0] - s * d[2], d[1], s * d[0] + c * d[2]]
    h = inner(ro, rd, t_min, t_max)
    if h is None:
        return None
    t, p, n, front = h
    p = [c * p[0] + s * p[2], p[1], -s * p[0] + c * p[2]]              # :131-136 (front_face kept)
    n = [c * n[0] + s * n[2], n[1], -s * n[0] + c * n[2]]
    p = add(p, off)                                                    # Translate: HitRecord::new again with the moved ray
    nn, front = hit_record_new(d, n)
    return {"t": bits(t), "normal": vbits(nn), "front_face": int(front), "p": vbits(p)}


def sphere_inner(center, radius):
    def f(o, d, t_min, t_max):
        c, r = vec(center), F(radius)
        oc = sub(o, c)
        a, half_b = dot(d, d), dot(oc, d)
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
        nn, front = hit_record_new(d, n)
        return root, p, nn, front
    return f


def rect_inner(kind, a0, a1, b0, b1, k):
    A, B, AX = {"xy": (0, 1, 2), "xz": (0, 2, 1), "yz": (1, 2, 0)}[kind]
    def f(o, d, t_min, t_max):
        a0_, a1_, b0_, b1_, k_ = F(a0), F(a1), F(b0), F(b1), F(k)
        with np.errstate(all="ignore"):
            t = (k_ - o[AX]) / d[AX]
        if t < t_min or t > t_max:
            return None
        a, b = o[A] + t * d[A], o[B] + t * d[B]
        if a < a0_ or a > a1_ or b < b0_ or b > b1_:
            return None
        n = [F(0)] * 3
        n[AX] = F(1)
        nn, front = hit_record_new(d, n)
        return t, at(o, d, t), nn, front
    return f


def logf_(x):
    """include/rt_portable_math.h logf_ (Cephes logf; this backend's ln, used by oracle and kernels instead of libm)"""
    x = F(x)
    b = int(np.array(x, np.float32).view(np.uint32))
    e = (b >> 23) - 126
    x = np.array((b & 0x007fffff) | 0x3f000000, np.uint32).view(np.float32)
    if x < F(0.707106781186547524):
        e -= 1
        x = x + x - F(1)
    else:
        x = x - F(1)
    z = x * x
    y = F(7.0376836292E-2)
    for cf in (-1.1514610310E-1, 1.1676998740E-1, -1.2420140846E-1, 1.4249322787E-1, -1.6668057665E-1, 2.0000714765E-1,
               -2.4999993993E-1, 3.3333331174E-1):
        y = y * x + F(cf)
    y = y * x * z
    fe = F(e)
    y = y + (F(-2.12194440e-4) * fe)
    y = y + (F(-0.5) * z)
    z = x + y
    return z + F(0.693359375) * fe


def reflect(v, n):
    """vec3.rs:740-753"""
    k = F(2) * dot(v, n)
    return sub(v, scale(k, n))


def refract_v(uv, n, eta):
    cos_theta = dot(neg(uv), n)
    par = scale(eta, add(uv, scale(cos_theta, n)))
    with np.errstate(all="ignore"):
        k = -np.sqrt(F(1) - dot(par, par))
    return add(par, scale(k, n))


def schlick_f(cosine, ior):
    r0 = (F(1) - ior) / (F(1) + ior)
    r0 = r0 * r0
    x = F(1) - cosine
    x2 = x * x
    return r0 + (F(1) - r0) * (x * (x2 * x2))


def onb_axes(n):
    w = normalize(n)
    a = vec([0, 1, 0]) if abs(w[0]) > F(0.9) else vec([1, 0, 0])
    v = normalize(cross(w, a))
    return cross(w, v), v, w


def metal_bounce(albedo, fuzz, d, normal, xi):
    """metal.rs:20-37 + types.rs:87-104; xi = the uniform numbers drawn (three per try of random_in_unit_sphere)"""
    d, n, fuzz = vec(d), vec(normal), F(fuzz)
    reflected = reflect(d, normalize(n))
    used = 0
    while True:
        p = [F(xi[used + i]) * (F(1) - F(-1)) + F(-1) for i in range(3)]     # random_real_range(-1, 1) = u * (hi - lo) + lo
        used += 3
        if dot(p, p) >= 1:
            continue
        break
    scattered = add(reflected, scale(fuzz, p))
    if not dot(scattered, n) > 0:
        return {"kind": 1, "used": used}
    return {"kind": 2, "dir": vbits(scattered), "attenuation": vbits(vec(albedo)), "used": used}


def dielectric_bounce(ior, d, normal, front_face, xi):
    """dielectric.rs:16-68"""
    d, n, ior = vec(d), vec(normal), F(ior)
    eta = F(1) / ior if front_face else ior
    uv = normalize(d)
    c = dot(neg(uv), n)
    cos_theta = c if c < 1 else F(1)                                          # f32::min
    sin_theta = np.sqrt(F(1) - cos_theta * cos_theta)
    if eta * sin_theta > 1:
        return {"kind": 2, "dir": vbits(reflect(uv, normalize(n))), "attenuation": vbits(vec([1, 1, 1])), "used": 0}
    if F(xi[0]) < schlick_f(cos_theta, eta):
        out = reflect(uv, n)
    else:
        out = refract_v(uv, n, eta)
    return {"kind": 2, "dir": vbits(out), "attenuation": vbits(vec([1, 1, 1])), "used": 1}


def rect_pdf_value(kind, a0, a1, b0, b1, k, origin, v):
    """rectangles.rs:54-67 (and the XZ / YZ twins): hit with (0.001, f32::MAX), then distance^2 / (cosine * area)"""
    h = rect_inner(kind, a0, a1, b0, b1, k)(vec(origin), vec(v), F(0.001), F(np.finfo(np.float32).max))
    if h is None:
        return F(0)
    t, _, nn, _ = h
    v = vec(v)
    area = (F(a1) - F(a0)) * (F(b1) - F(b0))
    distance_squared = t * t * dot(v, v)
    cosine = abs(dot(v, nn) / length(v))
    return distance_squared / (cosine * area)


def sphere_pdf_value(center, radius, origin, v):
    """sphere.rs:91-105"""
    h = sphere_inner(center, radius)(vec(origin), vec(v), F(0.001), F(np.finfo(np.float32).max))
    if h is None:
        return F(0)
    c, r, o = vec(center), F(radius), vec(origin)
    co = sub(c, o)
    cos_theta_max = np.sqrt(F(1) - r * r / dot(co, co))
    solid_angle = F(2) * F(np.pi) * (F(1) - cos_theta_max)
    return F(1) / solid_angle


def cosine_pdf_value(normal, direction):
    """pdf.rs:27-35"""
    _, _, w = onb_axes(vec(normal))
    cosine = dot(normalize(vec(direction)), w)
    return F(0) if cosine <= 0 else cosine * F(0.318309886183790671537767526745028724)


def lambertian_bounce(p, normal, lights, xi):
    """camera.rs:268-300 with lambertian.rs:32-52, pdf.rs:57-79, hittable_list.rs:58-76: the scattered direction, the mixture
    pdf value after the clamp and the scattering pdf.  lights = list of ("rect", kind, a0, a1, b0, b1, k) / ("sphere", c, r).
    Cosine sampling is only used with r1 = 0 (phi = 0: cos = 1, sin = 0 exactly), the sphere light is never sampled."""
    p, n = vec(p), vec(normal)
    used = 1
    if F(xi[0]) < F(0.5):                                                      # MixturePdf::generate: lights first
        k = int(F(xi[1]) * F(len(lights)))                                     # (this backend's random_int for a given number)
        k = min(k, len(lights) - 1)
        used = 2
        L = lights[k]
        assert L[0] == "rect"
        _, kind, a0, a1, b0, b1, kk = L
        A, B, AX = {"xy": (0, 1, 2), "xz": (0, 2, 1), "yz": (1, 2, 0)}[kind]
        pt = [F(0)] * 3
        pt[A] = F(xi[2]) * (F(a1) - F(a0)) + F(a0)
        pt[B] = F(xi[3]) * (F(b1) - F(b0)) + F(b0)
        pt[AX] = F(kk)
        used = 4
        direction = sub(pt, p)                                                 # rectangles.rs:69-77
    else:
        r1, r2 = F(xi[1]), F(xi[2])
        assert r1 == 0
        used = 3
        z = np.sqrt(F(1) - r2)
        local = [F(1) * np.sqrt(r2), F(0) * np.sqrt(r2), z]                    # types.rs:141-151 at phi = 0
        u, v, w = onb_axes(n)
        direction = add(add(scale(local[0], u), scale(local[1], v)), scale(local[2], w))
    weight = F(1) / F(len(lights))
    lsum = F(0)
    for L in lights:                                                           # hittable_list.rs:58-68
        val = rect_pdf_value(*L[1:], p, direction) if L[0] == "rect" else sphere_pdf_value(L[1], L[2], p, direction)
        lsum = lsum + weight * val
    pdf_val = F(0.5) * lsum + F(0.5) * cosine_pdf_value(n, direction)
    if abs(pdf_val) < F(1.0E-5):
        pdf_val = F(1.0E-4) if not np.signbit(pdf_val) else F(-1.0E-4)
    cosine = dot(n, normalize(direction))
    spdf = F(0) if cosine < 0 else cosine / F(np.pi)
    return {"kind": 3, "dir": vbits(direction), "pdf": bits(pdf_val), "scattering_pdf": bits(spdf), "used": used}


def medium_hit(center, radius, density, o, d, t_min, t_max, xi):
    """constant_medium.rs:46-94 over a sphere boundary; returns t or None"""
    big = F(np.finfo(np.float32).max)
    o, d = vec(o), vec(d)
    h1 = sphere_inner(center, radius)(o, d, -big, big)
    if h1 is None:
        return None
    h2 = sphere_inner(center, radius)(o, d, h1[0] + F(0.0001), big)
    if h2 is None:
        return None
    t1, t2 = h1[0], h2[0]
    if t1 < F(t_min):
        t1 = F(t_min)
    if t2 > F(t_max):
        t2 = F(t_max)
    if t1 >= t2:
        return None
    if t1 < 0:
        t1 = F(0)
    ray_length = length(d)
    inside = (t2 - t1) * ray_length
    neg_inv_density = F(-1) / F(density)
    hit_distance = neg_inv_density * logf_(xi)
    if hit_distance > inside:
        return None
    return t1 + hit_distance / ray_length


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: what VERDICT r2 still found on one reading: Hyperboloid (hyperboloid.rs:26-170, with the `+ 2` of its first phi),
# the multi-jittered sampler table with its shuffles and the concentric disk map (sampling.rs:26-149, 230-253) from the seed
# schedule's draws, the checker (checker_texture.rs:36-46), Camera::new and the four projections (camera.rs:39-184), and
# Cylinder::pdf_value / random (cylinder.rs:134-161).  Integer arithmetic in Python ints, floats in numpy f32 one operation at a time.

M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def mix32(h):                      # include/rt_seed_schedule.h: Murmur3's 32-bit finaliser
    h ^= h >> 16; h = (h * 0x85ebca6b) & M32; h ^= h >> 13; h = (h * 0xc2b2ae35) & M32; h ^= h >> 16
    return h


def hash3(a, b, c):
    h = mix32((a ^ 0x9e3779b9) & M32)
    h = mix32((h ^ ((b + 0x7f4a7c15) & M32)) & M32)
    return mix32((h ^ ((c + 0x165667b1) & M32)) & M32)


class Pcg:                         # PCG32 XSH-RR 64/32 (O'Neill), the seed schedule's generator
    def __init__(self, init_state, stream):
        self.state, self.inc = 0, ((stream << 1) | 1) & M64
        self.next_u32()
        self.state = (self.state + init_state) & M64
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & M64
        x = (((old >> 18) ^ old) >> 27) & M32
        rot = old >> 59
        return ((x >> rot) | (x << ((32 - rot) & 31))) & M32

    def real(self):                # types.rs:48-50 as the schedule draws it: 24 bits
        return F(self.next_u32() >> 8) * F(1.0 / 16777216.0)

    def int(self, lo, hi):         # types.rs:106-108, inclusive
        return lo + ((self.next_u32() * (hi - lo + 1)) >> 32)


def host_rng(seed, purpose):
    return Pcg(((hash3(seed, purpose, 0x5eed) << 32) | hash3(purpose, seed, 1)) & M64, purpose)


def sampler_table(spp, sets, seed):
    """SamplerBase::<MultiJittered>::new (sampling.rs:30-107 + 233-253) with the schedule's draws: points, the x and the y
    shuffle, the index table (one vector shuffled again for every set: Fisher-Yates from the end, the build's stand-in for
    rand's SliceRandom), the concentric map as (r, phi * 0.25) — sic, 0.25 not pi / 4 (:99)."""
    rng = host_rng(seed, 0x5a3b1e00)
    n = int(np.floor(np.sqrt(spp)))
    sub = F(1) / F(spp)
    pts = []
    for _ in range(sets):
        k = 0
        for i in range(n):
            for j in range(n):
                x = F(i * n + j) * sub + rng.real() * sub
                y = F(j * n + i) * sub + rng.real() * sub
                pts.append([x, y]); k += 1
        while k < spp:             # the build's extension for spp that are not squares: the rest on the sub-cell diagonal
            x = F(k) * sub + rng.real() * sub
            y = F(k) * sub + rng.real() * sub
            pts.append([x, y]); k += 1
    for c in (0, 1):
        for p in range(sets):
            for i in range(spp - 1):
                target = rng.int(0, spp - 1) + p * spp
                src = i + p * spp + 1
                pts[src][c], pts[target][c] = pts[target][c], pts[src][c]
    idx, shuffled = list(range(spp)), []
    for _ in range(sets):
        for i in range(spp - 1, 0, -1):
            j = rng.int(0, i)
            idx[i], idx[j] = idx[j], idx[i]
        shuffled += idx
    disk = []
    for x, y in pts:
        sx, sy = F(2) * x - F(1), F(2) * y - F(1)
        if sx > -sy:
            r, phi = (sx, sy / sx) if sx > sy else (sy, F(2) - sx / sy)
        elif sx < sy:
            r, phi = -sx, F(2) * F(2) + sy / sx
        else:
            r, phi = -sy, (F(6) - sx / sy if sy != 0 else F(0))
        disk.append([r, phi * F(0.25)])
    return pts, shuffled, disk


def sampler_index(seed, pixel, slot, sets, spp, shuffled):      # rt_seed_schedule.h: the sampler's state as a function of (pixel, slot)
    epoch = 1 if slot >= spp else 0
    rem = slot - (spp if epoch else 0)
    st = (hash3(seed ^ 0x51ed270b, pixel, epoch) * sets) >> 32
    return st * spp + shuffled[st * spp + rem]


def camera_new(look_from, look_at, up, vfov, aspect, aperture, focus):   # camera.rs:39-80
    look_from, look_at, up = vec(look_from), vec(look_at), vec(up)
    theta = (F(vfov) * F(np.pi)) / F(180)
    hh = F(np.tan(theta / F(2)))
    hw = F(aspect) * hh
    w = normalize(sub(look_from, look_at))
    u = normalize(cross(up, w))
    v = cross(w, u)
    f = F(focus)
    fv, fu, fw = scale(f, v), scale(f, u), scale(f, w)
    llc = sub(sub(sub(look_from, scale(hh, fv)), scale(hw, fu)), fw)
    return {"origin": look_from, "lower_left_corner": llc, "horizontal": scale(F(2), scale(hw, fu)), "vertical": scale(F(2), scale(hh, fv)),
            "u": u, "v": v, "w": w, "lens_radius": F(aperture) * F(0.5)}


def camera_ray(proj, cam, s, t, disk_xy, psi_max, lambda_max):
    """get_ray_perspective / _ortho / _fisheye / _spherical_panoramic (camera.rs:82-184) -> (origin, direction) or None"""
    one, two = F(1), F(2)
    if proj in ("perspective", "orthographic"):
        rdx, rdy = cam["lens_radius"] * disk_xy[0], cam["lens_radius"] * disk_xy[1]
        offset = add(scale(rdx, cam["u"]), scale(rdy, cam["v"]))
        on_plane = add(add(cam["lower_left_corner"], scale(s, cam["horizontal"])), scale(t, cam["vertical"]))
        if proj == "perspective":
            return add(cam["origin"], offset), sub(sub(on_plane, cam["origin"]), offset)
        return add(on_plane, offset), neg(cam["w"])
    pnx, pny = two * s - one, two * t - one
    if proj == "fisheye":
        rsqr = pnx * pnx + pny * pny
        if not rsqr <= one:
            return None
        r = np.sqrt(rsqr)
        psi = r * F(psi_max)
        sp, cp = F(np.sin(psi)), F(np.cos(psi))
        ca, sa = pnx / r, pny / r
        return cam["origin"], sub(add(scale(sp * ca, cam["u"]), scale(sp * sa, cam["v"])), scale(cp, cam["w"]))
    lam, psi = pnx * F(lambda_max), pny * F(psi_max)
    phi, theta = F(np.pi) - lam, F(0.5) * F(np.pi) - psi
    st_, ct = F(np.sin(theta)), F(np.cos(theta))
    sp, cp = F(np.sin(phi)), F(np.cos(phi))
    return cam["origin"], add(add(scale(st_ * sp, cam["u"]), scale(ct, cam["v"])), scale(st_ * cp, cam["w"]))


def checker_is_odd(freq, p):       # checker_texture.rs:37-45
    f = F(freq)
    sines = F(np.sin(f * p[0])) * F(np.sin(f * p[1])) * F(np.sin(f * p[2]))
    return bool(sines < 0)


def hyperboloid_new(p1, p2):       # hyperboloid.rs:26-67 -> p1, p2 (possibly swapped), zmin, zmax, ah, ch
    p1, p2 = vec(p1), vec(p2)
    zmin, zmax = min(p1[2], p2[2]), max(p1[2], p2[2])
    if p2[2] == 0:
        p1, p2 = p2, p1
    pp = list(p1)
    with np.errstate(all="ignore"):
        while True:
            pp = add(pp, scale(F(2), sub(p2, p1)))
            xy1 = pp[0] * pp[0] + pp[1] * pp[1]
            xy2 = p2[0] * p2[0] + p2[1] * p2[1]
            ah = (F(1) / xy1 - (pp[2] * pp[2]) / (xy1 * p2[2] * p2[2])) / (F(1) - (xy2 * pp[2] * pp[2]) / (xy1 * p2[2] * p2[2]))
            ch = (ah * xy2 - F(1)) / (p2[2] * p2[2])
            if not np.isinf(ah) and not np.isnan(ah):
                break
    return p1, p2, zmin, zmax, ah, ch


def hyperboloid_hit(p1, p2, phi_max, o, d, t_max):
    """hyperboloid.rs:78-169 (t_min is ignored, :78).  The normal goes through sin / cos of phi, so it is compared with a
    tolerance; t, v and front_face are exact.  Returns also which attempt gave the hit (the first adds 2 to a negative phi, :116)."""
    p1, p2, zmin, zmax, ah, ch = hyperboloid_new(p1, p2)
    o, d, t_max, phi_max = vec(o), vec(d), F(t_max), F(phi_max)
    two, one, zero = F(2), F(1), F(0)
    with np.errstate(all="ignore"):
        a = ah * d[0] * d[0] + ah * d[1] * d[1] - ch * d[2] * d[2]
        b = two * (ah * d[0] * o[0] + ah * d[1] * o[1] - ch * d[2] * o[2])
        c = ah * o[0] * o[0] + ah * o[1] * o[1] - ch * o[2] * o[2] - one
        n, (t0, t1) = poly_quadratic(a, b, c)
        if n == 0 or t0 > t_max or t1 <= zero:
            return None
        thit = t0
        if thit <= zero:
            thit = t1
            if thit > t_max:
                return None

        def inverse(t, wrap):
            ph = at(o, d, t)
            v = (ph[2] - p1[2]) / (p2[2] - p1[2])
            pr = add(scale(one - v, p1), scale(v, p2))
            phi = F(np.arctan2(pr[0] * ph[1] - ph[0] * pr[1], ph[0] * pr[0] + ph[1] * pr[1]))
            wrapped.append(bool(phi < zero))
            if phi < zero:
                phi = phi + wrap
            return ph, v, phi
        attempt, wrapped = 1, []
        ph, v, phi = inverse(thit, two)
        if ph[2] < zmin or ph[2] > zmax or phi > phi_max:
            if thit == t1:
                return None
            thit = t1
            if t1 > t_max:
                return None
            attempt = 2
            ph, v, phi = inverse(thit, F(np.pi) * two)
            if ph[2] < zmin or ph[2] > zmax or phi > phi_max:
                return None
        sn, cs = F(np.sin(phi)), F(np.cos(phi))
        dpdu = [-phi_max * ph[1], phi_max * ph[0], zero]
        dpdv = [(p2[0] - p1[0]) * cs - (p2[1] - p1[1]) * sn, (p2[0] - p1[0]) * sn + (p2[1] - p1[1]) * cs, p2[2] - p1[2]]
        n_out = normalize(cross(dpdu, dpdv))
        front = dot(d, n_out) < 0
        nn = n_out if front else neg(n_out)
    return {"t": bits(thit), "v": bits(v), "front_face": int(front), "normal": [float(x) for x in nn], "phi": float(phi), "attempt": attempt,
            "phi_was_negative": wrapped}


def cylinder_pdf_value(radius, zmin, zmax, phi_max, origin, v):
    """cylinder.rs:134-150: hit(Ray(o, v, 0), 0.0001, inf) -> (1 / area) / (|n . -v| / |o - p|^2), infinite -> 0"""
    h = quadric_hit("cylinder", [radius, zmin, zmax, phi_max], None, origin, v, 0.0001, np.inf)
    if h is None:
        return F(0)
    unb = lambda xs: [np.array(x, np.uint32).view(np.float32) for x in xs]
    p, n = unb(h["p"]), unb(h["normal"])
    origin, v = vec(origin), vec(v)
    area = (F(zmax) - F(zmin)) * F(radius) * F(phi_max)
    with np.errstate(all="ignore"):
        pdf = (F(1) / area) / (np.abs(dot(n, neg(v))) / dot(sub(origin, p), sub(origin, p)))
    return F(0) if np.isinf(pdf) else pdf


def cylinder_random(radius, zmin, zmax, origin, r1, r2):
    """cylinder.rs:152-161 + types.rs:153-163 + onb.rs:32-47 with given random numbers; sin / cos via numpy: compared with a tolerance"""
    centre = scale(F(0.5), add([-F(radius), -F(radius), F(zmin)], [F(radius), F(radius), F(zmax)]))
    direction = sub(centre, vec(origin))
    d2 = dot(direction, direction)
    axes = onb_axes(direction)
    r1, r2 = F(r1), F(r2)
    z = F(1) + r2 * (np.sqrt(F(1) - F(radius) * F(radius) / d2) - F(1))
    phi = F(2.0 * np.pi) * r1
    x, y = F(np.cos(phi)) * np.sqrt(F(1) - z * z), F(np.sin(phi)) * np.sqrt(F(1) - z * z)
    return add(add(scale(x, axes[0]), scale(y, axes[1])), scale(z, axes[2]))


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

    # ---- round 2 ----------------------------------------------------------------------------------------------------
    tri_v = [[0.0, 0.0, 5.0, 0.0, 0.2, -1.0, 0.0, 0.0], [2.0, 0.25, 5.5, 0.1, 0.0, -1.0, 1.0, 0.0], [0.5, 2.0, 4.5, 0.0, -0.1, -1.0, 0.25, 1.0]]
    tris = []
    for o, d, t_min, t_max in [
        ([0.7, 0.6, 0.0], [0.02, 0.01, 1.0], 0.001, np.inf),           # front side (normals point to -z): hit
        ([0.7, 0.6, 10.0], [0.02, 0.01, -1.0], 0.001, np.inf),         # from behind: culled by dot(d, n) > 0
        ([0.7, 0.6, 0.0], [0.02, 0.01, 1.0], 0.001, 4.0),              # beyond t_max
        ([3.0, 3.0, 0.0], [0.0, 0.0, 1.0], 0.001, np.inf),             # outside the triangle (inside the mesh box)
        ([0.0, 0.0, 0.0], [0.0, 0.0, 1.0], 0.001, np.inf),             # through vertex 1: b1 = b2 = 0 is inside
        ([0.7, 0.6, 0.0], [1.0, 0.125, 0.25], 0.001, np.inf),          # in the triangle's plane: div == 0 exactly
        ([1.1, 0.9, -3.0], [-0.05, 0.02, 1.0], 0.001, np.inf),
    ]:
        tris.append({"verts": tri_v, "o": o, "d": d, "t_min": t_min, "t_max": "inf" if t_max == np.inf else t_max,
                     "hit": mesh_hit(tri_v, o, d, t_min, t_max), "triangle_alone": tri_hit(tri_v, o, d, t_min, t_max)})
    flat = [[0.0, 0.0, 5.0, 0, 0, -1, 0, 0], [1.0, 0.0, 5.0, 0, 0, -1, 1, 0], [0.0, 1.0, 5.0, 0, 0, -1, 0, 1]]
    out["triangle"] = tris
    # a mesh that lies in an axis plane has a box of zero thickness, which Aabb::hit never reports (tmax <= tmin): never hit
    out["triangle_flat_mesh"] = {"verts": flat, "o": [0.2, 0.2, 0.0], "d": [0.0, 0.0, 1.0], "t_min": 0.001, "t_max": "inf",
                                 "hit": mesh_hit(flat, [0.2, 0.2, 0.0], [0.0, 0.0, 1.0], 0.001, np.inf),
                                 "triangle_alone": tri_hit(flat, [0.2, 0.2, 0.0], [0.0, 0.0, 1.0], 0.001, np.inf)}
    mov = []
    for c0, c1, t0, t1, r, o, d, time in [
        ([0, 0, 0], [0, 0.5, 0], 0.0, 1.0, 0.2, [0.05, 0.1, -3.0], [0.0, 0.02, 1.0], 0.0),
        ([0, 0, 0], [0, 0.5, 0], 0.0, 1.0, 0.2, [0.05, 0.1, -3.0], [0.0, 0.02, 1.0], 0.37),
        ([0, 0, 0], [0, 0.5, 0], 0.0, 1.0, 0.2, [0.05, 0.1, -3.0], [0.0, 0.02, 1.0], 1.0),      # centre has moved out of the ray
        ([1, 0.2, -2], [1.3, 0.6, -2.2], 0.25, 0.75, 0.3, [0.0, 0.0, 3.0], [0.3, 0.12, -1.5], 0.6),
        ([1, 0.2, -2], [1.3, 0.6, -2.2], 0.25, 0.75, 0.3, [0.0, 0.0, 3.0], [0.3, 0.12, -1.5], 2.5),   # a time outside the shutter (isotropic.rs:32)
    ]:
        mov.append({"c0": c0, "c1": c1, "time0": t0, "time1": t1, "radius": r, "o": o, "d": d, "time": time,
                    "hit": moving_sphere_hit(c0, c1, t0, t1, r, o, d, time, 0.001, np.inf)})
    out["moving_sphere"] = mov
    sin15, cos15 = float(F(0.25881904)), float(F(0.9659258))                 # any pair works: the wrapper only multiplies by them
    wraps = []
    for inner_desc, sincos, off, o, d in [
        (("sphere", [0, 0, 0], 1.0), (sin15, cos15), [2, 0.5, -1], [2.2, 0.4, -6.0], [0.01, 0.02, 1.0]),
        (("sphere", [0.5, 0, 0.25], 0.75), (float(F(-0.30901699)), float(F(0.95105652))), [-1, 0, 3], [0.0, 0.3, -4.0], [-0.1, -0.03, 1.0]),
        (("rect", "xy", 0, 165, 0, 330, 165), (sin15, cos15), [265, 0, 295], [278, 278, -800], [0.05, -0.2, 1.0]),   # Cornell's tall box, front side
        (("rect", "yz", 0, 330, 0, 165, 0), (sin15, cos15), [265, 0, 295], [278, 278, -800], [0.02, -0.25, 1.0]),
        (("rect", "xz", 0, 165, 0, 165, 330), (sin15, cos15), [265, 0, 295], [300, 540, 300], [0.1, -1.0, 0.2]),     # its top, from above
        (("rect", "xy", 0, 165, 0, 330, 165), (sin15, cos15), [265, 0, 295], [278, 278, -800], [0.9, -0.2, 1.0]),    # miss
    ]:
        if inner_desc[0] == "sphere":
            inner = sphere_inner(inner_desc[1], inner_desc[2])
        else:
            inner = rect_inner(*inner_desc[1:])
        h = wrapped_hit(inner, sincos[0], sincos[1], off, o, d, F(0.001), F(np.inf))
        wraps.append({"inner": list(inner_desc), "sin": sincos[0], "cos": sincos[1], "offset": off, "o": o, "d": d, "hit": h})
    out["wrapped"] = wraps
    # one bounce of ray_color with given random numbers
    bounces = []
    def sphere_rec(center, radius, o, d):
        h = sphere_inner(center, radius)(vec(o), vec(d), F(0.001), F(np.inf))
        return h                                                               # t, p, normal (face-forwarded), front
    for albedo, fuzz, o, d, xi in [
        ([0.8, 0.6, 0.2], 0.0, [0, 0, -3], [0.1, 0.2, 1.0], [0.5, 0.5, 0.5]),
        ([0.8, 0.6, 0.2], 0.3, [0, 0, -3], [0.1, 0.2, 1.0], [0.99, 0.99, 0.99, 0.25, 0.75, 0.5]),          # first try rejected (|p|^2 >= 1)
        ([0.7, 0.7, 0.7], 1.0, [0, 0.95, -3], [0.0, 0.0, 1.0], [0.5, 0.1, 0.5]),                            # grazing + fuzz pushes it below the surface: absorbed
    ]:
        t, p, n, front = sphere_rec([0, 0, 0], 1.0, o, d)
        e = metal_bounce(albedo, fuzz, d, n, xi)
        bounces.append({"material": "metal", "albedo": albedo, "fuzz": fuzz, "o": o, "d": d, "xi": xi, "t": bits(t), "expect": e})
    for ior, o, d, xi in [
        (1.5, [0, 0, -3], [0.1, 0.2, 1.0], [0.9]),            # front face, refracts
        (1.5, [0, 0, -3], [0.1, 0.2, 1.0], [0.0]),            # front face, Schlick reflection
        (1.5, [0, 0, 0.2], [0.3, 0.1, 1.0], [0.9]),           # from inside: back face, eta = 1.5, refracts out
        (1.5, [0.0, 0.9, 0.0], [1.0, 0.05, 0.0], [0.9]),      # from inside at a grazing angle: total internal reflection, no number drawn
    ]:
        t, p, n, front = sphere_rec([0, 0, 0], 1.0, o, d)
        e = dielectric_bounce(ior, d, n, front, xi)
        bounces.append({"material": "dielectric", "ior": ior, "o": o, "d": d, "xi": xi, "t": bits(t), "expect": e})
    lights = [("rect", "xz", 213, 343, 227, 332, 554), ("sphere", [190, 90, 190], 90.0)]                  # scene_cornell_box's `lights`
    for o, d, xi in [
        ([278, 278, -800], [0.0, -0.3, 1.0], [0.25, 0.1, 0.5, 0.5]),       # floor hit; light sampling, rectangle, its centre
        ([278, 278, -800], [0.0, -0.3, 1.0], [0.25, 0.49, 0.0, 1.0 - 2 ** -24]),   # a corner of the light
        ([278, 278, -800], [0.0, -0.3, 1.0], [0.75, 0.0, 0.5]),            # cosine sampling at phi = 0
        ([278, 278, -800], [0.0, -0.3, 1.0], [0.75, 0.0, 1.0 - 2 ** -24]), # cosine sampling, grazing: z -> 2^-12
        ([278, 278, -800], [0.0, -0.3, 1.0], [0.75, 0.0, 0.0]),            # straight up the normal: hits the light rectangle? (x, z decide)
        ([278, 278, -800], [0.0, -0.3, 1.0], [0.75, 0.0, 1.0]),            # r2 = 1 (the reference's f64 -> f32 cast can give it): z = 0, in the
                                                                           # surface, both pdfs 0 -> the clamp of camera.rs:279-288 makes it 1e-4
    ]:
        h = rect_inner("xz", 0, 555, 0, 555, 0)(vec(o), vec(d), F(0.001), F(np.inf))      # Cornell floor
        t, p, n, front = h
        e = lambertian_bounce(p, n, lights, xi)
        bounces.append({"material": "lambertian", "o": o, "d": d, "xi": xi, "t": bits(t), "expect": e})
    out["bounce"] = bounces
    meds = []
    for o, d, xi, t_max in [
        ([0, 0, -5], [0, 0, 1.0], 0.5, np.inf), ([0, 0, -5], [0, 0, 1.0], 0.999, np.inf), ([0, 0, -5], [0, 0, 1.0], 1e-30, np.inf),
        ([0, 0.5, 0], [0.3, 0.1, 1.0], 0.25, np.inf),       # origin inside the boundary: rec1.t < 0 is clamped to t_min
        ([0, 0, -5], [0, 0.1, 2.0], 0.5, np.inf),           # direction not unit length
        ([0, 3, -5], [0, 0, 1.0], 0.5, np.inf),             # misses the boundary
    ]:
        t = medium_hit([0, 0, 0], 2.0, 0.2, o, d, 0.001, t_max, xi)
        meds.append({"center": [0, 0, 0], "radius": 2.0, "density": 0.2, "o": o, "d": d, "xi": xi, "t": None if t is None else bits(t)})
    out["medium"] = meds
    # the reference's own known answers next to the path: math/src/mat4.rs:657-667 and math/src/colors.rs:589-629
    out["mat4_transpose"] = {"m": list(range(16)), "result": [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]}
    out["colors"] = {"html": [["eb4034", [235, 64, 52, 255]], ["#eb4034", [235, 64, 52, 255]], ["#ffeb4034", [235, 64, 52, 255]],
                              ["ff34eb9e", [52, 235, 158, 255]], ["#ff34eb9e", [52, 235, 158, 255]], ["jdk4fxuu", [0, 0, 0, 0]]],
                     "to_u32": [[[52, 235, 158, 255], 0xff34eb9e]]}
    inv_cases = []
    for m in mats[1:]:
        inv, _ = mat4_inverse_and_normal([[F(e) for e in row] for row in m])
        inv_cases.append({"m": [x for row in m for x in row], "inverse": [bits(e) for row in inv for e in row]})
    out["mat4_invert"] = inv_cases

    # ---- round 3 ----------------------------------------------------------------------------------------------------
    seed = 0x5EED
    tables = []
    for spp in (4, 9, 10):
        pts, shuffled, disk = sampler_table(spp, 5, seed)
        tables.append({"spp": spp, "sets": 5, "seed": seed, "samples": [vbits(p) for p in pts], "shuffled": shuffled,
                       "disk_r_phi": [vbits(d) for d in disk]})
    out["sampler"] = tables
    out["checker"] = [{"freq": f, "p": p, "odd": checker_is_odd(f, vec(p))} for f, p in (
        (4.0, [0.3, 0.2, 0.1]), (4.0, [-0.3, 0.2, 0.1]), (10.0, [13.2, -1000.0, 3.7]), (0.32, [278.0, 554.0, 279.5]), (4.0, [1.0, 1.0, 1.0]),
        (4.0, [-2.5, 0.01, 7.75]), (3.0, [0.5, -0.5, -0.5]))]
    # one frame per projection: an XY rectangle that emits a checker, seen by 1 sample per pixel at depth 1, so a pixel's sum is the
    # odd colour, the even colour or the background: which one follows from the table, the camera and the ray above alone
    W, H, spp, sets = 40, 24, 4, 83
    pts, shuffled, disk = sampler_table(spp, sets, seed)
    rect = dict(kind="xy", a0=-3.0, a1=2.5, b0=-1.5, b1=2.0, k=-1.0)
    frames = []
    to_radians = lambda deg: F(deg) * (F(np.pi) / F(180))        # f32::to_radians (main.rs:1199-1200)
    for proj, psi_deg, lambda_deg, aperture in (("perspective", 0, 0, 0.4), ("orthographic", 0, 0, 0.4), ("fisheye", 70.0, 0, 0.0),
                                                ("spherical_panoramic", 50.0, 80.0, 0.0)):
        psi_max, lambda_max = to_radians(psi_deg), to_radians(lambda_deg)
        cam = camera_new([0.7, 0.4, 5.0], [0.1, 0.2, -1.0], [0.0, 1.0, 0.0], 50.0, W / H, aperture, 6.0)
        classes, rays = [], []
        for y in range(H):
            for x in range(W):
                pixel = y * W + x
                lens = proj in ("perspective", "orthographic")
                off = pts[sampler_index(seed, pixel, 0, sets, spp, shuffled)]
                u = (F(x) + off[0]) / F(W - 1)
                v = F(1) - (F(y) + off[1]) / F(H - 1)
                dk = [F(0), F(0)]
                if lens:
                    r_, phi_ = disk[sampler_index(seed, pixel, 1, sets, spp, shuffled)]
                    dk = [r_ * F(np.cos(phi_)), r_ * F(np.sin(phi_))]
                ray = camera_ray(proj, cam, u, v, dk, psi_max, lambda_max)
                cls = 0            # 0 background (or no ray), 1 odd, 2 even, 3 the back of the emitter (black)
                if ray is not None:
                    o, d = ray
                    with np.errstate(all="ignore"):
                        h = rect_hit(rect["kind"], rect["a0"], rect["a1"], rect["b0"], rect["b1"], rect["k"], [float(c) for c in o],
                                     [float(c) for c in d], 0.001, np.inf)
                    if h is not None:
                        t = np.array(h["t"], np.uint32).view(np.float32)
                        cls = 3 if not h["front_face"] else (1 if checker_is_odd(2.0, at(o, d, t)) else 2)
                    if (x, y) in ((0, 0), (W // 2, H // 2), (W - 1, H - 1), (7, 19)):
                        rays.append({"x": x, "y": y, "u": bits(u), "v": bits(v), "disk": vbits(dk), "origin": vbits(o), "direction": vbits(d)})
                classes.append(cls)
        frames.append({"projection": proj, "psi_max_degrees": psi_deg, "lambda_max_degrees": lambda_deg, "width": W, "height": H, "spp": spp, "sets": sets,
                       "seed": seed, "camera": {k: (bits(v) if np.ndim(v) == 0 else vbits(v)) for k, v in cam.items()},
                       "look": {"from": [0.7, 0.4, 5.0], "at": [0.1, 0.2, -1.0], "up": [0.0, 1.0, 0.0], "vfov": 50.0, "aspect": W / H,
                                "aperture": aperture, "focus": 6.0},
                       "rect": rect, "checker_freq": 2.0, "classes": "".join(str(c) for c in classes), "rays": rays})
    out["projection_frames"] = frames
    hyp = []
    two_pi = float(F(np.pi) * F(2))
    for p1, p2, phi_max, o, d, t_max in [
        ([1.0, 0.0, -1.0], [0.6, 0.8, 1.0], two_pi, [4.0, 0.3, 0.2], [-1.0, 0.0, 0.05], np.inf),
        ([1.0, 0.0, -1.0], [0.6, 0.8, 1.0], two_pi, [0.0, 0.0, 0.0], [1.0, 0.2, 0.1], np.inf),          # from inside
        ([1.0, 0.0, -1.0], [0.6, 0.8, 1.0], two_pi, [4.0, 0.3, 5.0], [-1.0, 0.0, 0.05], np.inf),         # passes above zmax: miss
        ([1.0, 0.0, -1.0], [0.6, 0.8, 1.0], two_pi, [4.0, 0.3, 0.2], [-1.0, 0.0, 0.05], 2.0),            # beyond t_max
        ([1.0, 0.0, -1.0], [0.6, 0.8, 1.0], 3.0, [4.0, -0.6, 0.2], [-1.0, 0.05, 0.0], np.inf),           # phi < 0 on the first attempt: + 2 (sic)
        ([1.0, 0.0, -1.0], [0.6, 0.8, 1.0], 1.0, [4.0, -0.6, 0.2], [-1.0, 0.05, 0.0], np.inf),           # ... clipped, second root
        ([0.5, 0.5, 0.0], [1.0, -0.2, 2.0], two_pi, [3.0, 2.0, 1.0], [-1.0, -0.7, 0.0], np.inf),         # p2.z != 0, p1.z == 0: no swap
        ([1.0, -0.2, 2.0], [0.5, 0.5, 0.0], two_pi, [3.0, 2.0, 1.0], [-1.0, -0.7, 0.0], np.inf),         # p2.z == 0: swapped (:34)
        ([0.5, 0.5, 0.0], [1.0, -0.2, 2.0], two_pi, [-3.0, 0.1, 1.5], [1.0, 0.02, -0.1], np.inf),
    ]:
        hyp.append({"p1": p1, "p2": p2, "phi_max": phi_max, "o": o, "d": d, "t_max": "inf" if t_max == np.inf else t_max,
                    "coefficients": vbits(hyperboloid_new(p1, p2)[4:6]), "hit": hyperboloid_hit(p1, p2, phi_max, o, d, t_max)})
    # ... and rays found by search (fixed seed) for the branches the hand-made ones do not reach: a negative phi on the first
    # attempt (+ 2, sic, :116), on the second (+ 2 pi, :138), a first root clipped away by phi_max or z (second attempt, :121-143)
    rng3 = np.random.default_rng(2026)
    want = {"neg_first": 3, "second": 3, "neg_second": 2}
    while any(want.values()):
        p1 = [1.0, 0.0, -1.0]; p2 = [0.6, 0.8, 1.0]
        phi_max = float(F(rng3.choice([two_pi, 4.5, 3.0, 1.5])))
        o = [float(F(x)) for x in rng3.normal(size=3) * 3.0]
        tgt = rng3.normal(size=3) * 0.6
        d = [float(F(tgt[i] - o[i])) for i in range(3)]
        h = hyperboloid_hit(p1, p2, phi_max, o, d, np.inf)
        if h is None:
            continue
        kind = None
        if h["attempt"] == 1 and h["phi_was_negative"][0]: kind = "neg_first"
        elif h["attempt"] == 2 and h["phi_was_negative"][1]: kind = "neg_second"
        elif h["attempt"] == 2: kind = "second"
        if kind is None or want[kind] == 0:
            continue
        if abs(h["phi"] - phi_max) < 1e-3:
            continue               # the clip must not hang on the last bits of an atan2
        want[kind] -= 1
        hyp.append({"p1": p1, "p2": p2, "phi_max": phi_max, "o": o, "d": d, "t_max": "inf", "coefficients": vbits(hyperboloid_new(p1, p2)[4:6]), "hit": h})
    out["hyperboloid"] = hyp
    cyl = []
    for radius, zmin, zmax, phi_max, origin, v in [
        (0.5, -1.0, 1.0, two_pi, [0.0, 3.0, 0.0], [0.0, -1.0, 0.0]), (0.5, -1.0, 1.0, two_pi, [2.0, 3.0, 0.4], [-0.6, -1.0, -0.1]),
        (0.5, -1.0, 1.0, two_pi, [0.0, 3.0, 5.0], [0.0, -1.0, 0.0]), (0.3, -0.5, 0.5, 4.0, [0.0, 3.0, 0.0], [0.0, -1.0, 0.0]),
        (0.8, 0.0, 2.0, two_pi, [0.1, 0.2, 1.0], [1.0, 0.3, 0.2]),          # from inside the tube
    ]:
        cyl.append({"radius": radius, "zmin": zmin, "zmax": zmax, "phi_max": phi_max, "origin": origin, "v": v,
                    "pdf": bits(cylinder_pdf_value(radius, zmin, zmax, phi_max, origin, v))})
    out["cylinder_pdf"] = cyl
    out["cylinder_random"] = [{"radius": r_, "zmin": z0, "zmax": z1, "origin": o, "r1": r1, "r2": r2,
                               "direction": [float(c) for c in cylinder_random(r_, z0, z1, o, r1, r2)]} for r_, z0, z1, o, r1, r2 in (
        (0.5, -1.0, 1.0, [0.0, 3.0, 0.0], 0.25, 0.5), (0.5, -1.0, 1.0, [4.0, 0.2, -3.0], 0.7, 0.1), (0.8, 0.0, 2.0, [-2.0, 2.0, 5.0], 0.01, 0.99))]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
