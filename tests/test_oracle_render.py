"""Properties of the oracle's renderer that need no GPU: determinism, additivity over sample
ranges, BVH == flat list, energy conservation (furnace), libm vs portable math at image level."""
import numpy as np
import pytest

import oracle_ffi as O

rt = O.rt


@pytest.fixture(scope="module")
def cornell():
    scene, params, camera, sampler = rt.configs.build("C2", image_width=48, samples_per_pixel=16)
    return scene, params, camera, sampler, O.OracleScene(scene)


def test_deterministic_and_thread_independent(cornell):
    scene, params, camera, sampler, o = cornell
    a, ra = o.render(params, camera, sampler, threads=1)
    b, rb = o.render(params, camera, sampler, threads=8)
    assert ra == rb and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c, _ = o.render(params, camera, sampler, seed=1)
    assert not np.array_equal(a, c)


def test_sample_ranges_add_up(cornell):
    scene, params, camera, sampler, o = cornell
    full, rays = o.render(params, camera, sampler)
    a, ra = o.render(params, camera, sampler, spp_begin=0, spp_count=5)
    b, rb = o.render(params, camera, sampler, spp_begin=5, spp_count=11)
    assert ra + rb == rays
    np.testing.assert_allclose(a + b, full, rtol=1e-5, atol=1e-6)
    assert (full[..., 3] == 16).all()


def test_tiles_match_the_full_frame(cornell):
    scene, params, camera, sampler, o = cornell
    full, _ = o.render(params, camera, sampler)
    tile, _ = o.render(params, camera, sampler, tile=(8, 16, 40, 33))
    assert np.array_equal(tile[16:33, 8:40].view(np.uint32), full[16:33, 8:40].view(np.uint32))
    assert not tile[:16].any() and not tile[:, :8].any()


def test_reference_bvh_equals_flat_list_on_book2():
    scene, params, camera, sampler = rt.configs.build("C3", image_width=64, samples_per_pixel=4)
    rng = np.random.default_rng(5)
    n = 20000
    rays = np.zeros((7, n), np.float32)
    rays[0:3] = np.array([[478], [278], [-600]], np.float32) + rng.normal(0, 5, (3, n)).astype(np.float32)
    rays[3:6] = (np.array([[-200], [0], [600]], np.float32) + rng.normal(0, 250, (3, n))).astype(np.float32)
    rays[6] = rng.random(n)
    bvh = O.OracleScene(scene, bvh_seed=1).first_hit(rays)
    bvh2 = O.OracleScene(scene, bvh_seed=99).first_hit(rays)
    flat = O.OracleScene(scene, exhaustive=True).first_hit(rays)
    assert (bvh[0] != 0xFFFFFFFF).mean() > 0.3
    for k in range(5):
        assert np.array_equal(bvh[k], flat[k]) and np.array_equal(bvh2[k], flat[k])


def test_furnace_white_sphere_in_white_background():
    """A white Lambertian sphere under uniform white light must stay white: the estimator
    emitted + albedo * scattering_pdf * L / pdf has expectation 1 for every pixel."""
    s = rt.host.Scene()
    white = s.lambertian(s.solid(1, 1, 1))
    ball = s.sphere((0, 0, 0), 1, white)
    light = s.xz_rect(-1, 1, -1, 1, 5, s.diffuse_light(s.solid(0, 0, 0)))
    s.set_roots(s.hittable_list([ball]), s.hittable_list([light]))
    cfg = rt.host.RaytracerConfig.load()
    p = cfg.scene_params("CornellBox").replace(image_width=24, aspect_ratio=1.0, samples_per_pixel=1024, look_from=(0, 0, 4),
                                               look_at=(0, 0, 0), vertical_fov=40.0, background=(1, 1, 1), aperture=0.0)
    sampler = rt.host.MultiJitteredSampler(1024)
    img, _ = O.OracleScene(s).render(p, p.camera(), sampler)
    mean = img[..., :3].sum() / (24 * 24 * 3 * 1024)
    assert abs(mean - 1.0) < 0.02, mean


def test_libm_and_portable_math_agree_at_image_level():
    scene, params, camera, sampler = rt.configs.build("C1", image_width=96, samples_per_pixel=16)
    a, _ = O.OracleScene(scene).render(params, camera, sampler)
    b, _ = O.OracleScene(scene, use_libm=True).render(params, camera, sampler)
    ra, rb = O.resolve(a, 16)[..., :3], O.resolve(b, 16)[..., :3]
    rmse = float(np.sqrt(np.mean((ra - rb) ** 2)))
    assert rmse < 0.02, rmse          # a few paths diverge after an ulp-level difference; the image does not


def test_resolve_rules():
    s = np.array([[4.0, 0.0, np.nan, 7], [np.inf, 1e-40, 64.0, 1], [-4.0, 1.0, 0.25, 2]], np.float32)
    out = O.resolve(s, 4)
    assert out[0].tolist() == [1.0, 0.0, 0.0, 7.0]               # !is_normal (0, NaN) -> 0 ; alpha untouched
    assert out[1].tolist() == [0.0, 0.0, 1.0, 1.0]               # inf, subnormal -> 0 ; sqrt(16) clamped to 1
    assert np.isnan(out[2, 0]) and out[2, 1] == 0.5 and out[2, 2] == 0.25   # sqrt of a negative normal sum stays NaN (main.rs:1366)
