// rt_host_io.cpp — the output step after the render path: the screenshot file and resumable
// accumulation.
//
//   PNG      main.rs:1543-1556 saves `raytraced_<time>.png` through image 0.24.6 / png 0.17.8 (Cargo.lock; neither is under
//            /root/reference).  Here: an 8-bit RGB, non-interlaced PNG with filter 0 on every scanline, deflated by zlib —
//            any PNG reader gives back the same pixels, which is all the format promises (the byte stream of the
//            reference's encoder is not reproduced).
//   RTCK     no counterpart in the reference (it can only start over): the un-normalised sums of rt_render plus how far
//            the sample range got.  Because rt_render over [a, b) then [b, c) gives the bits of [a, c)
//            (tests/test_gpu_pipeline.py), a render resumed from a checkpoint equals the uninterrupted one bit for bit.
#include <zlib.h>

#include <cerrno>

#include "rt_host_internal.h"

using namespace rth;

namespace {

void put_u32(std::vector<uint8_t> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }

void chunk(std::vector<uint8_t> &png, const char type[4], const uint8_t *data, size_t n) {
    put_u32(png, (uint32_t)n);
    size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put_u32(png, (uint32_t)crc32(0L, png.data() + at, (uInt)(n + 4)));
}

struct CheckpointHeader {   // little-endian, 40 bytes
    char magic[4];          // "RTCK"
    uint32_t version;       // 1
    int32_t width, height;
    int32_t samples_per_pixel, spp_done;
    uint32_t seed;
    uint32_t sums_crc32;    // of the width * height * 4 floats that follow
    uint64_t n_floats;
};

}  // namespace

extern "C" {

int rth_write_png_rgb8(const char *path, int32_t width, int32_t height, const uint8_t *rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) { set_error("png: bad arguments"); return RT_E_INVALID; }
    const size_t row = (size_t)width * 3;
    std::vector<uint8_t> raw((row + 1) * (size_t)height);
    for (int32_t y = 0; y < height; ++y) {
        raw[(row + 1) * (size_t)y] = 0;   // filter type None
        memcpy(&raw[(row + 1) * (size_t)y + 1], rgb + row * (size_t)y, row);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_error("png: deflate failed"); return RT_E_INVALID; }
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, (uint32_t)width); put_u32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8 /* bit depth */, 2 /* colour type RGB */, 0, 0, 0 /* no interlace */};
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(png, "IHDR", ihdr.data(), ihdr.size());
    chunk(png, "IDAT", z.data(), zn);
    chunk(png, "IEND", nullptr, 0);
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("png: cannot open %s: %s", path, strerror(errno)); return RT_E_INVALID; }
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    if (fclose(f) != 0 || !ok) { set_error("png: short write to %s", path); return RT_E_INVALID; }
    return RT_OK;
}

int rth_checkpoint_save(const char *path, int32_t width, int32_t height, int32_t samples_per_pixel, int32_t spp_done, uint32_t seed,
                        const float *rgba_sum) {
    if (!path || !rgba_sum || width <= 0 || height <= 0 || spp_done < 0 || spp_done > samples_per_pixel) { set_error("checkpoint: bad arguments"); return RT_E_INVALID; }
    CheckpointHeader h{};
    memcpy(h.magic, "RTCK", 4);
    h.version = 1; h.width = width; h.height = height; h.samples_per_pixel = samples_per_pixel; h.spp_done = spp_done; h.seed = seed;
    h.n_floats = (uint64_t)width * (uint64_t)height * 4u;
    uLong crc = crc32(0L, Z_NULL, 0);
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(rgba_sum);
    for (uint64_t off = 0, total = h.n_floats * 4; off < total;) {   // crc32 takes 32-bit lengths
        const uInt n = (uInt)std::min<uint64_t>(total - off, 1u << 30);
        crc = crc32(crc, bytes + off, n);
        off += n;
    }
    h.sums_crc32 = (uint32_t)crc;
    std::string tmp = std::string(path) + ".tmp";   // written aside and renamed: an interrupted save leaves the old file
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { set_error("checkpoint: cannot open %s: %s", tmp.c_str(), strerror(errno)); return RT_E_INVALID; }
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(rgba_sum, 4, (size_t)h.n_floats, f) == (size_t)h.n_floats;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); set_error("checkpoint: write to %s failed", path); return RT_E_INVALID; }
    return RT_OK;
}

int rth_checkpoint_load(const char *path, int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed, float *rgba_sum,
                        int32_t *spp_done) {
    if (!path || !rgba_sum || !spp_done) { set_error("checkpoint: bad arguments"); return RT_E_INVALID; }
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("checkpoint: cannot open %s: %s", path, strerror(errno)); return RT_E_INVALID; }
    CheckpointHeader h{};
    bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "RTCK", 4) == 0 && h.version == 1;
    if (ok && (h.width != width || h.height != height || h.samples_per_pixel != samples_per_pixel || h.seed != seed ||
               h.n_floats != (uint64_t)width * (uint64_t)height * 4u || h.spp_done < 0 || h.spp_done > samples_per_pixel)) {
        fclose(f);
        set_error("checkpoint: %s belongs to another render (%dx%d, %d spp, seed %u)", path, h.width, h.height, h.samples_per_pixel, h.seed);
        return RT_E_INVALID;
    }
    ok = ok && fread(rgba_sum, 4, (size_t)h.n_floats, f) == (size_t)h.n_floats;
    fclose(f);
    if (!ok) { set_error("checkpoint: %s is not a complete RTCK file", path); return RT_E_INVALID; }
    uLong crc = crc32(0L, Z_NULL, 0);
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(rgba_sum);
    for (uint64_t off = 0, total = h.n_floats * 4; off < total;) {
        const uInt n = (uInt)std::min<uint64_t>(total - off, 1u << 30);
        crc = crc32(crc, bytes + off, n);
        off += n;
    }
    if ((uint32_t)crc != h.sums_crc32) { set_error("checkpoint: %s is corrupt (checksum)", path); return RT_E_INVALID; }
    *spp_done = h.spp_done;
    return RT_OK;
}

}  // extern "C"
