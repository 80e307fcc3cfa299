"""Input step: texture files to RGBA8 by the host library's own decoder (csrc/rt_host_image.cpp),
the step ImageTexture::new does with the `image` crate (image_texture.rs:20-38).

PNG is lossless: the decoder must give PIL's bytes exactly.  JPEG is not: the decoder restates jpeg-decoder 0.3.0 (fixed-point
IDCT, triangle-filter chroma upsampling, 20-bit colour conversion), PIL is libjpeg-turbo (another IDCT, another upsampling
rounding) — PARITY UNPINNED by the reference; the bound tested here is "the same image to within a few LSB"."""
import io
import json
import os
import struct

import numpy as np
import pytest
from PIL import Image

import oracle_ffi as O

rt = O.rt
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
SHIPPED = ["textures/misc/earthmap.jpg"] + ["textures/uv_grids/ash_uvgrid%02d.jpg" % i for i in (1, 3, 8, 9, 10)]


def pil_rgba(src):
    return np.asarray(Image.open(src).convert("RGBA"))


def diff(a, b):
    assert a.shape == b.shape and a.dtype == np.uint8
    return np.abs(a.astype(np.int32) - b.astype(np.int32))


@pytest.mark.parametrize("rel", SHIPPED)
def test_shipped_jpegs_match_libjpeg_within_a_few_lsb(rel):
    path = os.path.join(DATA, rel)
    got = rt.host.load_image_rgba(path)
    want = pil_rgba(path)
    d = diff(got, want)
    assert (got[..., 3] == 255).all()
    assert d.max() <= 4
    if rel.endswith("10.jpg"):          # 4:2:0: the two decoders round the chroma interpolation differently
        assert d.mean() < 0.15 and (d > 1).mean() < 0.04
    else:
        assert d.mean() < 0.01 and (d > 1).mean() < 0.001
    if rel.endswith("08.jpg"):          # greyscale file -> Luma to Rgba (r = g = b)
        assert (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all()


def encode(img, **kw):
    buf = io.BytesIO()
    img.save(buf, **kw)
    return buf.getvalue()


@pytest.mark.parametrize("size", [(64, 48), (37, 21), (1, 1), (17, 8), (8, 33)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])           # 4:4:4, 4:2:2, 4:2:0
def test_synthetic_jpegs_ragged_sizes_and_restarts(size, subsampling):
    rng = np.random.default_rng(size[0] * 100 + size[1] + subsampling)
    # smooth content + noise: exercises all coefficients without living on the clamp
    yy, xx = np.mgrid[0:size[1], 0:size[0]]
    base = np.stack([128 + 100 * np.sin(xx / 5.0), 128 + 100 * np.cos(yy / 7.0), 128 + 60 * np.sin((xx + yy) / 3.0)], -1)
    img = Image.fromarray(np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8), "RGB")
    for kw in (dict(quality=92), dict(quality=35), dict(quality=92, restart_marker_blocks=1), dict(quality=80, restart_marker_rows=1)):
        data = encode(img, format="JPEG", subsampling=subsampling, **kw)
        got = rt.host.decode_image_rgba(data)
        d = diff(got, pil_rgba(io.BytesIO(data)))
        assert got.shape == (size[1], size[0], 4)
        # 4:4:4 differs only by IDCT and colour rounding; subsampled chroma adds the interpolation rounding (and, at the
        # 1-pixel-wide right edge of odd widths, a different edge rule)
        assert d.max() <= (4 if subsampling == 0 else 24), (kw, d.max())
        assert d.mean() < (0.15 if subsampling == 0 else 1.5), (kw, d.mean())


def test_flat_grey_jpeg_is_exact():
    # DC-only blocks: the IDCT has nothing to round, Cb = Cr = 128 leaves the conversion exact -> every decoder agrees
    for v in (0, 1, 127, 128, 200, 255):
        data = encode(Image.new("RGB", (24, 16), (v, v, v)), format="JPEG", quality=100, subsampling=2)
        got = rt.host.decode_image_rgba(data)
        want = pil_rgba(io.BytesIO(data))
        assert np.array_equal(got, want)
        assert (got[..., :3] == got[0, 0, 0]).all() and abs(int(got[0, 0, 0]) - v) <= 1


def test_greyscale_and_adobe_rgb_jpegs():
    rng = np.random.default_rng(5)
    g = Image.fromarray(rng.integers(0, 256, (19, 30), dtype=np.uint8), "L")
    data = encode(g, format="JPEG", quality=90)
    got = rt.host.decode_image_rgba(data)
    assert diff(got, pil_rgba(io.BytesIO(data))).max() <= 2
    # an Adobe APP14 segment with transform = 0 declares the three components to be RGB, not YCbCr: splice one into a file
    # whose components PIL wrote as YCbCr and expect the raw planes back (grey input: Y = v, Cb = Cr = 128)
    data = encode(Image.new("RGB", (16, 16), (90, 90, 90)), format="JPEG", quality=100, subsampling=0)
    app14 = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
    spliced = data[:2] + app14 + data[2:]
    got = rt.host.decode_image_rgba(spliced)
    assert (got[..., 0] == 90).all() and (got[..., 1] == 128).all() and (got[..., 2] == 128).all()
    ycc = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x01"      # transform = 1: YCbCr, as the shipped uv grids say
    assert (rt.host.decode_image_rgba(data[:2] + ycc + data[2:])[..., :3] == 90).all()


def png_cases():
    rng = np.random.default_rng(11)
    h, w = 23, 31
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgb[:, 5:20] = (rgb[:, 5:20] // 64) * 64          # runs, so that the encoder picks different filters per line
    yield "rgb", Image.fromarray(rgb, "RGB")
    yield "rgba", Image.fromarray(np.dstack([rgb, rng.integers(0, 256, (h, w), dtype=np.uint8)]), "RGBA")
    yield "grey", Image.fromarray(rgb[..., 0], "L")
    yield "grey_alpha", Image.fromarray(rgb[..., :2].copy(), "LA")
    yield "palette", Image.fromarray(rgb, "RGB").quantize(64)
    yield "palette_4bit", Image.fromarray(rgb, "RGB").quantize(13)
    yield "palette_1bit", Image.fromarray(rgb, "RGB").quantize(2)
    yield "bilevel", Image.fromarray(rgb[..., 0] > 128)
    big = np.add.outer(np.arange(200), np.arange(300)).astype(np.uint8)   # gradients: Sub / Up / Average / Paeth all in use
    yield "gradient", Image.fromarray(np.dstack([big, big.T[:200, :300] if False else big[::-1], big // 2]), "RGB")


@pytest.mark.parametrize("name,img", list(png_cases()))
def test_png_is_exact(name, img):
    data = encode(img, format="PNG", optimize=(name != "rgb"))
    got = rt.host.decode_image_rgba(data)
    assert np.array_equal(got, pil_rgba(io.BytesIO(data)))


def test_png_palette_transparency_and_16_bit():
    pal = Image.new("P", (4, 2))
    pal.putpalette([10, 20, 30, 40, 50, 60, 70, 80, 90] + [0] * (768 - 9))
    pal.putdata([0, 1, 2, 0, 1, 2, 0, 1])
    data = encode(pal, format="PNG", transparency=bytes([255, 0, 128]))
    got = rt.host.decode_image_rgba(data)
    assert got[0, 0].tolist() == [10, 20, 30, 255] and got[0, 1].tolist() == [40, 50, 60, 0] and got[0, 2].tolist() == [70, 80, 90, 128]
    g16 = np.array([[0, 255, 256, 65535], [4660, 32768, 65280, 511]], np.uint16)
    data = encode(Image.fromarray(g16, "I;16"), format="PNG")
    got = rt.host.decode_image_rgba(data)
    assert np.array_equal(got[..., 0], (g16 >> 8).astype(np.uint8)) and (got[..., 3] == 255).all()


def glb_images(path):
    raw = open(path, "rb").read()
    magic, _, _ = struct.unpack_from("<III", raw, 0)
    assert magic == 0x46546C67
    jlen, _ = struct.unpack_from("<II", raw, 12)
    doc = json.loads(raw[20:20 + jlen])
    bin_at = 20 + jlen + 8
    for im in doc.get("images", []):
        bv = doc["bufferViews"][im["bufferView"]]
        at = bin_at + bv.get("byteOffset", 0)
        yield im.get("mimeType"), raw[at:at + bv["byteLength"]]


def test_images_embedded_in_the_teapot_glb():
    # geometry_import.rs:461-488 decodes these (the materials they feed are then ignored, triangle_mesh.rs:238-239)
    seen = 0
    for mime, data in glb_images(os.path.join(DATA, "models/teapot/teapot.glb")):
        got = rt.host.decode_image_rgba(data)
        want = pil_rgba(io.BytesIO(data))
        if mime == "image/png":
            assert np.array_equal(got, want)
        else:
            assert diff(got, want).max() <= 4
        seen += 1
    assert seen == 3


def test_errors_are_codes_not_crashes(tmp_path):
    H = rt.host
    Err = rt._ffi.RtError
    with pytest.raises(Err):
        H.load_image_rgba(str(tmp_path / "missing.jpg"))
    with pytest.raises(Err, match="neither"):
        H.decode_image_rgba(b"GIF89a" + bytes(64))
    img = Image.fromarray(np.random.default_rng(2).integers(0, 256, (40, 40, 3), dtype=np.uint8), "RGB")
    with pytest.raises(Err, match="progressive"):
        H.decode_image_rgba(encode(img, format="JPEG", progressive=True))
    with pytest.raises(Err, match="CMYK"):
        H.decode_image_rgba(encode(img.convert("CMYK"), format="JPEG"))
    good = encode(img, format="JPEG", quality=90)
    for cut in (3, 20, 200, len(good) // 2):
        try:        # a cut inside the entropy-coded data decodes what is there (zeros after), a cut in the headers is an error
            H.decode_image_rgba(good[:cut])
        except Err:
            pass
    png = bytearray(encode(img, format="PNG"))
    png[60] ^= 0x55
    with pytest.raises(Err, match="CRC|inflate"):
        H.decode_image_rgba(bytes(png))
    with pytest.raises(Err):
        H.decode_image_rgba(encode(img, format="PNG")[:100])
    # random bytes behind a JPEG signature: an error or an image, never a crash
    rng = np.random.default_rng(3)
    for _ in range(200):
        junk = bytearray(good)
        for k in rng.integers(2, len(junk), 12):
            junk[k] = rng.integers(0, 256)
        try:
            H.decode_image_rgba(bytes(junk))
        except Err:
            pass


def test_assets_use_the_library_decoder(monkeypatch):
    # the asset path of the reference's scene builders must not need PIL
    import builtins
    real = builtins.__import__

    def no_pil(name, *a, **k):
        if name.split(".")[0] == "PIL":
            raise ImportError("PIL is not on the asset path")
        return real(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_pil)
    a = rt.host.Assets(need=("earthmap",))
    assert a.c.earthmap.width == 1024 and a.c.earthmap.height == 512
