"""bench.py — Mrays/s of the HIP path tracer on the BASELINE.json workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2] [--no-cpu]

A step = one full frame of the workload: every pixel, every sample, depth 50, accumulated into
an HBM-resident W*H*4 f32 buffer (scene, BVH and sampler tables are resident before the timed
region).  N=1: Cornell box 600x600 at 1000 spp (BASELINE.json configs[1]).  N>1: one process
per GPU; rank g renders samples [g*S, (g+1)*S) of an N*S-spp frame (S = 1000) and the per-GPU
sums are added on rank 0 by one RCCL reduce inside the timed region ("weak": per-GPU work is fixed).
A ray = one closest-hit query issued by ray_color (camera.rs:264).  The number of rays per step
is a constant of (scene, seed) and is counted once, untimed, by the counting kernel variant.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FLOPs per unit of work, SURVEY.md §8(d) (derived by counting the cited reference code).
F_BOX = 27          # per child-box test with 1/d hoisted (+3 divides per ray, below)
F_RAY = 3
F_SPHERE, F_MOVING_EXTRA, F_RECT, F_TRI = 30, 10, 12, 59
F_QUADRIC = 21 + 45  # world2object of the ray (2 x 3 rows x 7) + coefficients, roots, one point, atan2, clip (cone.rs:43-100)
F_ACCEPT = 18       # hit record of the accepted primitive (sphere figure; rect 14, tri 55 are close / rarer)
F_MEDIUM = 2 * 30 + 12
F_SHADE = [140, 45, 60, 2, 20]      # lambertian+mixture pdf, metal, dielectric, emitted, isotropic
F_TEX = [0, 8, 1000, 12]            # solid, checker, noise (7 octaves), image
F_CAMERA = 40
PEAK_FP32_VALU_TFLOPS = 157.3       # MI355X_MICROARCH.md "Peak FP32 (vector)"


def algorithmic_flops(st):
    pt = st["prim_tests"]
    f = st["box_tests"] * F_BOX + st["rays"] * F_RAY
    f += pt[0] * F_SPHERE + pt[1] * (F_SPHERE + F_MOVING_EXTRA) + (pt[2] + pt[3] + pt[4]) * F_RECT + pt[5] * F_TRI + sum(pt[6:]) * F_QUADRIC
    f += st["accepted_hits"] * F_ACCEPT + st["medium_tests"] * F_MEDIUM
    f += sum(n * c for n, c in zip(st["shades"], F_SHADE)) + sum(n * c for n, c in zip(st["tex_evals"], F_TEX))
    f += st["light_pdf_evals"] * 30 + st["paths"] * F_CAMERA
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--device-bvh", nargs="?", const="ploc", default=None, choices=["ploc", "lbvh"],
                    help="build the BVH on the GPU (rt_bvh_build.hip) instead of on the host: clustering (default) or Morton radix tree")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rtiow_loader
    rt = rtiow_loader.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1 or os.environ.get("RT_BENCH_FORCE_DIST"):   # the env switch rehearses the RCCL calls with one rank on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    over = {}
    if args.width:
        over["image_width"] = args.width
    base_spp = args.spp or rt.configs.WORKLOADS[args.workload]["samples_per_pixel"]
    total_spp = base_spp * world
    over["samples_per_pixel"] = total_spp
    scene, params, camera, sampler = rt.configs.build(args.workload, **over)
    W, H = params.image_width, params.image_height
    gpu = rt.backend.HipBackend(local_rank)   # raises if the HIP library or the GPU is missing
    t_flat = time.perf_counter()
    flat = scene.flatten(device_bvh=args.device_bvh is not None)
    t_flat = time.perf_counter() - t_flat
    if args.device_bvh == "lbvh" and flat.desc.build_bvh_on_device:
        flat.desc.build_bvh_on_device = 2      # RT_BVH_DEVICE_LBVH
    gpu.upload_scene(flat)
    info = gpu.scene_info()
    gpu.upload_sampler(sampler)
    spp_begin, spp_count = rank * base_spp, base_spp

    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream

    def step():
        acc.zero_()
        gpu.render_device(params, camera, acc.data_ptr(), sptr, spp_begin=spp_begin, spp_count=spp_count)
        if dist is not None:
            dist.reduce(acc, dst=0)   # RCCL over xGMI: per-GPU sums -> rank 0

    # rays per step: counted once by the counting variant (same paths, same seed), untimed
    host_img = np.zeros((H, W, 4), np.float32)
    _, st = gpu.render(params, camera, spp_begin=spp_begin, spp_count=spp_count, out=host_img, stats=True)
    rays_step = torch.tensor([st["rays"], st["paths"]], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(rays_step)
    total_rays, total_paths = float(rays_step[0].item()), float(rays_step[1].item())

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        acc.zero_()
        ev[k][0].record(stream)
        gpu.render_device(params, camera, acc.data_ptr(), sptr, spp_begin=spp_begin, spp_count=spp_count)
        ev[k][1].record(stream)
        if dist is not None:
            dist.reduce(acc, dst=0)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    if rank == 0:
        # sanity: the device image of the last step equals the counted host render (world 1) / is finite
        img = acc.cpu().numpy()
        if world == 1:
            assert np.array_equal(np.nan_to_num(img), np.nan_to_num(host_img)), "timed kernel and counting kernel disagree"
        flops = algorithmic_flops(st)
        # HBM-side bytes per step: PMC passes of this very workload (tools/pmc.sh), committed under profiles/
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "r1", "%s_%dspp_pmc.json" % (args.workload.lower(), base_spp))
        if world == 1 and not args.width and os.path.exists(prof):
            with open(prof) as f:
                ks = json.load(f)["kernels"].values()
            traffic = sum(2.0 * (k.get("fetch_size_kb") or 0.0) + (k.get("write_size_kb") or 0.0) for k in ks) * 1024.0
            traffic_src = os.path.relpath(prof, ROOT)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/sec", "value": total_rays * args.steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %s %dx%d, %d spp per GPU (%d total), depth %d" % (
                args.workload, rt.configs.WORKLOADS[args.workload]["scene"], W, H, base_spp, total_spp, params.max_ray_depth),
                "sharding": "sample ranges per GPU + RCCL reduce to rank 0" if world > 1 else "single GPU",
                "seed": rt.configs.SEED, "primitives": int(flat.desc.n_prims), "bvh_nodes": int(info["n_bvh_nodes"]),
                       "bvh": "built on the GPU (%s) in %.2f ms (host flatten %.2f s)" % (args.device_bvh, info["bvh_build_ms"], t_flat) if info["built_on_device"]
                              else "built on the host in %.2f s (flatten)" % t_flat},
            "rays_per_step": total_rays, "paths_per_step": total_paths, "node_visits_per_ray": round(st["node_visits"] / max(1, st["rays"]), 2),
            "mpaths_per_s": total_paths * args.steps / elapsed / 1e6,
            # lane occupancy per kernel phase (counting variant): lanes working / lanes issued
            "phase_occupancy": {n: (round(l / s_, 4) if s_ else None) for n, l, s_ in zip(
                ("node_step", "leaf_step", "shade", "path_start", "begin_trace"), st["phase_lanes"], st["phase_slots"])},
            "phase_wave_runs_per_ray": {n: round(s_ / 64.0 / max(st["rays"], 1) * 64, 3) for n, s_ in zip(
                ("node_step", "leaf_step", "shade", "path_start", "begin_trace"), st["phase_slots"])},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                         "kernel": "one rt_render call = the wavefront pipeline (rtk::wf_generate_kernel, wf_traverse_kernel, "
                                   "wf_shade_kernel, wf_tail_kernel, wf_fold_kernel; wf_shade_kernel dominates on this workload)",
                         "kernel_ms": kernel_ms, "algorithmic_flops_per_launch": flops,
                         "note": "no dense contraction -> FP32 vector roofline (SURVEY.md 8d); flops = work counters x per-unit constants; "
                                 "kernel_ms = HIP events on the caller's stream around the call (the pipelines' streams fork from and join it); "
                                 "traffic = FETCH_SIZE*2 + WRITE_SIZE bytes per step over all kernels, from " + (traffic_src or "n/a")},
        }
        if not args.no_cpu and world == 1:      # the CPU baseline is an N=1 figure; at N>1 the other ranks would only wait for it
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_ffi as O   # cpu_baseline leg only: the oracle is the checker/baseline, never the product
            cores = min(16, os.cpu_count() or 1)
            oracle = O.OracleScene(scene)
            c0 = time.perf_counter()
            _, cal_rays = oracle.render(params, camera, sampler, spp_begin=0, spp_count=1, threads=cores)   # calibration, untimed
            cal = time.perf_counter() - c0
            sample_spp = max(1, min(base_spp, int(15.0 / max(cal, 1e-3))))                                   # about 15 s of CPU work
            c0 = time.perf_counter()
            _, c_rays = oracle.render(params, camera, sampler, spp_begin=0, spp_count=sample_spp, threads=cores)
            c_dt = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": c_rays / c_dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                                   "sample": "%dx%d, samples 0..%d of %d, depth %d, %d threads, %.1f s" % (
                                       W, H, sample_spp, total_spp, params.max_ray_depth, cores, c_dt)}
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
