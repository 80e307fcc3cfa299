"""A whole frame of a BASELINE workload at a given spp, GPU against the oracle (test infrastructure: this tool is the checker, not
the product): ray total, per-pixel ray counts, NaN pixels, the sums.  Too long for the suite (the oracle needs a minute or two on
16 threads), so it is run by hand and its output kept under profiles/.  usage: python3 tests/fullframe_parity.py C3 100 [bands=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this file lives in tests/: the oracle is loaded by tests only)
import numpy as np
import rtiow_loader
import oracle_ffi as O

rt = rtiow_loader.load()
name, spp = sys.argv[1], int(sys.argv[2])
bands = int(sys.argv[3]) if len(sys.argv) > 3 else 8
scene, params, camera, sampler = rt.configs.build(name, samples_per_pixel=spp)
gpu = rt.backend.HipBackend(0)
gpu.upload_scene(scene.flatten())
gpu.upload_sampler(sampler)
W, H = params.image_width, params.image_height
t0 = time.time()
g_plain, _ = gpu.render(params, camera)
print("%s %dx%d, %d spp: GPU (product kernels) %.2f s" % (name, W, H, spp, time.time() - t0), flush=True)
g_img, stats, g_rays = gpu.render(params, camera, stats=True, want_pixel_rays=True)
print("counting variant: %d rays; same bits as the product kernels: %s" % (stats["rays"], np.array_equal(g_plain.view(np.uint32), g_img.view(np.uint32))), flush=True)
oracle = O.OracleScene(scene)
c_img = np.zeros_like(g_img); c_rays = np.zeros_like(g_rays); c_total = 0
t0 = time.time()
for b in range(bands):
    y0, y1 = H * b // bands, H * (b + 1) // bands
    img, total, pr = oracle.render(params, camera, sampler, tile=(0, y0, W, y1), want_pixel_rays=True, threads=16)
    c_img[y0:y1] = img[y0:y1]; c_rays[y0:y1] = pr[y0:y1]; c_total += total
    print("  oracle rows %d..%d: %d rays so far, %.0f s" % (y0, y1, c_total, time.time() - t0), flush=True)
bad = g_rays != c_rays
print("ray totals: gpu %d oracle %d; pixels with other ray counts: %d of %d" % (stats["rays"], c_total, bad.sum(), W * H))
# Every pixel with another ray count: which sample it is, and what the oracle says about that sample when its BvhNodes test every
# child instead of culling by their boxes (orc_options.exhaustive: the flat-list semantics the kernels' padded tree reproduces,
# DESIGN.md §3 "BVH conservativeness" — the reference's BvhNode can cull a grazing hit that the leaf's own test accepts).
def oracle_rays(tile, k):
    """The rays of the oracle's path for sample k of the pixel `tile`, in order (its stderr trace, one line per ray)."""
    import tempfile
    sys.stderr.flush()
    rays = []
    with tempfile.TemporaryFile(mode="w+b") as f:
        saved = os.dup(2)
        os.dup2(f.fileno(), 2)
        O.lib().orc_set_trace(1)
        try:
            oracle.render(params, camera, sampler, tile=tile, spp_begin=k, spp_count=1, threads=1)
        finally:
            O.lib().orc_set_trace(0)
            os.dup2(saved, 2); os.close(saved)
        f.seek(0)
        for line in f.read().decode(errors="replace").splitlines():
            w = line.split()
            if len(w) > 12 and w[0] == "[orc]" and w[1] == "depth":
                rays.append([float(w[4]), float(w[5]), float(w[6]), float(w[8]), float(w[9]), float(w[10]), float(w[12])])
    return rays


flat = O.OracleScene(scene, exhaustive=True) if bad.any() else None
explained = n_samples = 0
for y, x in np.argwhere(bad)[:200]:
    tile = (int(x), int(y), int(x) + 1, int(y) + 1)
    for k in range(spp):
        _, _, gr = gpu.render(params, camera, tile=tile, spp_begin=k, spp_count=1, stats=True, want_pixel_rays=True)
        c1, _, cr = oracle.render(params, camera, sampler, tile=tile, spp_begin=k, spp_count=1, want_pixel_rays=True, threads=1)
        if gr[y, x] == cr[y, x]:
            continue
        n_samples += 1
        g1, _ = gpu.render(params, camera, tile=tile, spp_begin=k, spp_count=1)
        f1, _, fr = flat.render(params, camera, sampler, tile=tile, spp_begin=k, spp_count=1, want_pixel_rays=True, threads=1)
        same = fr[y, x] == gr[y, x] and np.allclose(np.nan_to_num(f1[y, x, :3]), np.nan_to_num(g1[y, x, :3]), rtol=3e-4, atol=1e-6)
        note = " = gpu, same value: the reference's BvhNode culls a grazing hit by its box" if same else ""
        if not same:     # a tie (two leaves of one BvhNode at the same t): the reference keeps the right child (bvh.rs:83), and which leaf
            # is on the right depends on the node's random axes — another seed of those axes may decide it the kernels' way
            for seed in range(1, 9):
                o2 = O.OracleScene(scene, bvh_seed=seed)
                s1, _, sr = o2.render(params, camera, sampler, tile=tile, spp_begin=k, spp_count=1, want_pixel_rays=True, threads=1)
                if sr[y, x] == gr[y, x] and np.allclose(np.nan_to_num(s1[y, x, :3]), np.nan_to_num(g1[y, x, :3]), rtol=3e-4, atol=1e-6):
                    same = True
                    note = "; with BvhNode axes from seed %d the oracle has %d = gpu, same value: a tie between two leaves of one BvhNode" % (seed, sr[y, x])
                    break
        if not same:     # ... or no seed does (the two leaves are ordered the same way along every axis): look at the rays themselves.  Up
            # to the bounce at which they part, the oracle's path is the GPU's (same random numbers), and the oracle's trace has
            # its rays; both sides answer for each of them through their first-hit hooks: the same t to the last bit on two
            # different primitives is the tie
            for j, ray in enumerate(oracle_rays(tile, k)):
                r7 = np.array(ray, np.float32).reshape(7, 1)
                gh, ch = gpu.trace_first_hit(r7), oracle.first_hit(r7)
                if gh[0][0] == ch[0][0]:
                    continue
                if gh[1][:1].view(np.uint32)[0] == ch[1][:1].view(np.uint32)[0]:
                    same = True
                    note = "; ray %d of the path meets primitives %d (gpu: the later one in list order) and %d (oracle: its BvhNode's right child) at the same t = %.9g: a tie that no seed of the axes turns" % (j + 1, gh[0][0], ch[0][0], gh[1][0])
                else:
                    note = "  <-- ray %d of the path: gpu prim %d t %.9g, oracle prim %d t %.9g" % (j + 1, gh[0][0], gh[1][0], ch[0][0], ch[1][0])
                break
        explained += bool(same)
        print("   pixel x %d y %d sample %d: gpu %d rays, oracle %d, oracle without box culling %d%s" % (x, y, k, gr[y, x], cr[y, x], fr[y, x], note if (same or note) else "  <-- not explained"), flush=True)
if bad.any():
    print("of the %d pixels (%d looked at: %d differing samples), %d samples are explained by the reference's BvhNode (box culling of a grazing hit, or a tie between its leaves)" % (bad.sum(), min(int(bad.sum()), 200), n_samples, explained))
g_nan, c_nan = np.isnan(g_img[..., :3]).any(-1), np.isnan(c_img[..., :3]).any(-1)
print("NaN pixels: gpu %d oracle %d, the same ones: %s" % (g_nan.sum(), c_nan.sum(), np.array_equal(g_nan, c_nan)))
ok = ~(g_nan | c_nan)
rel = np.abs(g_img[ok][:, :3] - c_img[ok][:, :3]) / np.maximum(np.abs(c_img[ok][:, :3]), 1e-3)
print("sums: largest relative difference %.3g (tolerance of the suite 3e-4), bit-identical pixels %.4f %%" % (rel.max(), 100.0 * (g_img.view(np.uint32) == c_img.view(np.uint32)).all(-1).mean()))
g8 = (np.clip(gpu.resolve(g_img, spp)[..., :3], 0, 1) * 255).astype(np.int32); c8 = (np.clip(O.resolve(c_img, spp)[..., :3], 0, 1) * 255).astype(np.int32)
print("resolved 8-bit image: pixels differing by 1: %d, by more: %d" % ((np.abs(g8 - c8).max(-1) == 1).sum(), (np.abs(g8 - c8).max(-1) > 1).sum()))
gpu.close()
sys.exit(0 if (n_samples == explained and np.array_equal(g_nan, c_nan)) else 1)
