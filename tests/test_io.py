"""Output step: screenshot pixels + PNG file, resumable accumulation (csrc/rt_host_io.cpp, rt_resolve_rgb8)."""
import os

import numpy as np
import pytest

import oracle_ffi as O

rt = O.rt


def test_rgb8_conversion_known_answers():
    # main.rs:1363-1388 then image 0.24.6 to_rgb8: round(clamp(sqrt(sum / spp), 0, 1) * 255), halves away from zero
    spp = 4
    c = np.array([0.0, 0.5 / 255, 0.49 / 255, 1.5 / 255, 1.0, 2.0, 127.5 / 255], np.float64)     # the values wanted after the gamma pass
    sums = np.zeros((1, 8, 4), np.float32)
    sums[0, :7, 0] = (c * c * spp).astype(np.float32)
    sums[0, 7, 0] = np.nan
    sums[0, :, 1] = np.inf                                  # !is_normal -> 0 (main.rs:1369)
    sums[0, :, 2] = 1e-45                                   # subnormal -> 0
    got = O.resolve_rgb8(sums, spp)
    f = O.resolve(sums, spp)[0, :, 0].astype(np.float64) * 255.0
    want = np.floor(f + 0.5).astype(np.uint8)               # by construction of the cases no f + 0.5 is inexact
    assert got[0, :, 0].tolist() == want.tolist()
    assert got[0, 0, 0] == 0 and got[0, 4, 0] == 255 and got[0, 5, 0] == 255 and got[0, 7, 0] == 0
    assert not got[..., 1].any() and not got[..., 2].any()


def test_png_file_round_trip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    for shape in ((1, 1, 3), (37, 53, 3), (600, 800, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        path = str(tmp_path / "shot.png")
        rt.host.write_png(path, img)
        back = Image.open(path)
        assert back.mode == "RGB" and back.size == (shape[1], shape[0])
        assert np.array_equal(np.asarray(back), img)
    with pytest.raises(rt.RtError):
        rt.host.write_png(str(tmp_path / "no" / "such" / "dir.png"), img)


def test_checkpoint_round_trip_and_refusals(tmp_path):
    rng = np.random.default_rng(2)
    sums = rng.random((48, 64, 4), dtype=np.float32)
    sums[3, 5, 0] = np.nan                                  # NaN sums are legitimate (main.rs:1369 zeroes them later)
    path = str(tmp_path / "frame.rtck")
    rt.host.checkpoint_save(path, sums, 256, 96)
    back, done = rt.host.checkpoint_load(path, 64, 48, 256)
    assert done == 96 and np.array_equal(back.view(np.uint32), sums.view(np.uint32))
    rt.host.checkpoint_save(path, sums, 256, 128)           # overwrite = write aside + rename
    assert rt.host.checkpoint_load(path, 64, 48, 256)[1] == 128 and not os.path.exists(path + ".tmp")
    for bad in (dict(width=65), dict(samples_per_pixel=255), dict(seed=1)):
        kw = dict(width=64, height=48, samples_per_pixel=256, seed=0x5EED)
        kw.update(bad)
        with pytest.raises(rt.RtError):
            rt.host.checkpoint_load(path, **kw)
    raw = bytearray(open(path, "rb").read())
    raw[200] ^= 1                                           # one flipped bit in the sums
    open(path, "wb").write(raw)
    with pytest.raises(rt.RtError) as e:
        rt.host.checkpoint_load(path, 64, 48, 256)
    assert "corrupt" in str(e.value)
    open(path, "wb").write(raw[:1000])                      # truncated
    with pytest.raises(rt.RtError):
        rt.host.checkpoint_load(path, 64, 48, 256)
    with pytest.raises(rt.RtError):
        rt.host.checkpoint_save(path, sums, 256, 300)       # more samples done than the frame has


@pytest.mark.gpu
def test_rgb8_on_the_gpu_matches_the_oracle_and_resume_is_bit_exact(gpu, tmp_path):
    scene, params, camera, sampler = rt.configs.build("C2", image_width=96, samples_per_pixel=16)
    gpu.upload_scene(scene.flatten())
    gpu.upload_sampler(sampler)
    full, _ = gpu.render(params, camera)
    rgb = gpu.resolve_rgb8(full, sampler.spp)
    assert rgb.shape == (96, 96, 3) and np.array_equal(rgb, O.resolve_rgb8(full, sampler.spp))      # byte-exact
    edge = np.zeros((1, 6, 4), np.float32)
    edge[0, :, 0] = [0.0, np.nan, np.inf, 1e-45, 16.0 * (0.5 / 255) ** 2, 1e9]
    assert np.array_equal(gpu.resolve_rgb8(edge, 16), O.resolve_rgb8(edge, 16))
    # interrupted after 6 samples, resumed from the file in a fresh buffer
    path = str(tmp_path / "c2.rtck")
    part, _ = gpu.render(params, camera, spp_begin=0, spp_count=6)
    rt.host.checkpoint_save(path, part, sampler.spp, 6)
    sums, done = rt.host.checkpoint_load(path, params.image_width, params.image_height, sampler.spp)
    gpu.render(params, camera, spp_begin=done, spp_count=sampler.spp - done, out=sums)
    assert np.array_equal(sums.view(np.uint32), full.view(np.uint32))
    shot = str(tmp_path / "c2.png")
    rt.host.write_png(shot, rgb)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(shot)), rgb)
