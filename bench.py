"""bench.py — Mrays/s of the HIP path tracer on the BASELINE.json workloads.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C1] [--spp S] [--no-cpu] [--no-sub]

A step = one full frame of the workload: every pixel, every sample, depth 50, accumulated into an HBM-resident
W*H*4 f32 buffer (scene, BVH and sampler tables are resident before the timed region).

Headline (`value`, N=1): **C1**, the Book-1 final scene 1200x675 at 10 spp — the configuration BASELINE.json's target
(>= 100x the host CPU at 1 GPU) is stated on; its BVH walk is the path north_star is about.  The same JSON line carries
a `workloads` dict with the other single-GPU configurations at their FULL sizes — C2 Cornell box 600x600 at 1000 spp,
C3 Book-2 final scene 800x800 at 10 000 spp, C4 triangle-mesh scene 1920x1080 at 256 spp over 2.26 M triangles — each
with ms_per_step, Mrays/s, its roofline fraction and node_visits_per_ray, measured the same way (`--no-sub` skips them).

N>1: one process per GPU (the driver launches torch.distributed.run; run without a launcher, `--gpus N` spawns the N
ranks itself before anything touches a GPU).  Weak scaling: rank g renders samples [g*S, (g+1)*S) of an N*S-spp frame
(S = the workload's spp) and the per-GPU sums are added on rank 0 by one RCCL reduce inside the timed region.
`--strong` keeps the frame fixed instead (the workload's spp split over the ranks; BASELINE's C5 is such a frame).

A ray = one closest-hit query issued by ray_color (camera.rs:264).  The number of rays per step is a constant of
(scene, seed) and is counted once, untimed, by the counting kernel variant, which traces the identical paths.
"""
import argparse
import json
import os
import statistics
import subprocess
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
PROFILE_ROUND = "r3"
KERNELS = ("generate", "traverse", "shade", "tail")

# sub-workloads of the default run: (name, timed steps, warm-up steps).  C3's frame is 6.4e9 paths: one timed step.
SUB_WORKLOADS = (("C2", 3, 1), ("C3", 1, 0), ("C4", 3, 1))


def algorithmic_flops(st):
    """counters x constants: box tests are the child boxes actually tested (rt_stats.box_tests)."""
    pt = st["prim_tests"]
    f = st["box_tests"] * F_BOX + st["rays"] * F_RAY
    f += pt[0] * F_SPHERE + pt[1] * (F_SPHERE + F_MOVING_EXTRA) + (pt[2] + pt[3] + pt[4]) * F_RECT + pt[5] * F_TRI + sum(pt[6:]) * F_QUADRIC
    f += st["accepted_hits"] * F_ACCEPT + st["medium_tests"] * F_MEDIUM
    f += sum(n * c for n, c in zip(st["shades"], F_SHADE)) + sum(n * c for n, c in zip(st["tex_evals"], F_TEX))
    f += st["light_pdf_evals"] * 30 + st["paths"] * F_CAMERA
    return f


def kernel_busy_ms(name, base_spp):
    """Busy time per frame of every render kernel, from the committed rocprofv3 --kernel-trace --stats summary of this workload
    (profiles/<round>/<wl>_<spp>spp_kernel_stats.csv: the product kernels are the `<false, ...>` instantiations, the traced run
    renders steps + warm-up frames with them, profiles/<round>/<wl>_<spp>spp_ktrace_bench.json says how many)."""
    import csv
    base = os.path.join(ROOT, "profiles", PROFILE_ROUND, "%s_%dspp" % (name.lower(), base_spp))
    if not (os.path.exists(base + "_kernel_stats.csv") and os.path.exists(base + "_ktrace_bench.json")):
        return None, None
    with open(base + "_ktrace_bench.json") as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    frames = line.get("frames_rendered", line["steps"] + line["warmup"])
    busy = {}
    with open(base + "_kernel_stats.csv") as f:
        for r in csv.DictReader(f):
            n = r["Name"]
            for k in KERNELS + ("fold",):
                if "rtk::wf_%s_kernel" % k in n and ("<false" in n or k == "fold"):
                    # (the fold kernel is not a template: its rows hold the counting run's frame too)
                    per = frames + 1 if k == "fold" else frames
                    busy[k] = busy.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6 / per
    # ... and the time each kernel needs when it has the chip to itself: SQ_BUSY_CYCLES of the PMC passes, whose dispatches run one
    # at a time (tools/pmc_report.py: busy_ms)
    if os.path.exists(base + "_pmc.json"):
        with open(base + "_pmc.json") as f:
            for n, k in json.load(f)["kernels"].items():
                for key in KERNELS + ("fold",):
                    if "wf_%s_kernel" % key in n and k.get("busy_ms"):
                        busy[key + "_alone"] = k["busy_ms"]
    return busy, os.path.relpath(base + "_kernel_stats.csv", ROOT)


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start the N ranks as child processes (this process has not touched a GPU
    and never will) and exit with the worst of their codes.  Rank 0's JSON line is the children's stdout."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


class Bench:
    def __init__(self, args, rt, torch, dist, rank, local_rank, world):
        self.args, self.rt, self.torch, self.dist = args, rt, torch, dist
        self.rank, self.world = rank, world
        self.dev = torch.device("cuda", local_rank)
        self.gpu = rt.backend.HipBackend(local_rank)   # raises if the HIP library or the GPU is missing
        self.stream = torch.cuda.current_stream(self.dev)
        # the frame's one exchange goes through the C-ABI (rt_comm_reduce_to_root = RCCL reduce on the render's stream);
        # torch.distributed only carries the 128-byte communicator id, the barriers and the scalar sums of the report
        self.comm = None
        if dist is not None:
            self.comm = rt.sharding.Communicator(self.gpu, rank, world, exchange=lambda box: dist.broadcast_object_list(box, src=0))

    def run(self, name, steps, warmup, spp=0, width=0, cpu=False):
        """One workload: upload, count rays (untimed), warm up, time `steps` frames.  Returns the result dict."""
        args, rt, torch, dist, gpu, dev = self.args, self.rt, self.torch, self.dist, self.gpu, self.dev
        import numpy as np
        over = {}
        if width:
            over["image_width"] = width
        base_spp = spp or rt.configs.WORKLOADS[name]["samples_per_pixel"]
        if args.strong:
            total_spp = base_spp
            spp_begin, spp_count = rt.sharding.sample_range(self.rank, self.world, total_spp)
        else:
            total_spp = base_spp * self.world
            spp_begin, spp_count = self.rank * base_spp, base_spp
        over["samples_per_pixel"] = total_spp
        t_build = time.perf_counter()
        scene, params, camera, sampler = rt.configs.build(name, **over)
        t_build = time.perf_counter() - t_build
        W, H = params.image_width, params.image_height
        t_flat = time.perf_counter()
        flat = scene.flatten(device_bvh=args.device_bvh is not None)
        t_flat = time.perf_counter() - t_flat
        if args.device_bvh == "lbvh" and flat.desc.build_bvh_on_device:
            flat.desc.build_bvh_on_device = 2      # RT_BVH_DEVICE_LBVH
        gpu.upload_scene(flat)
        info = gpu.scene_info()
        gpu.upload_sampler(sampler)

        acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        sptr = self.stream.cuda_stream

        # rays per step: counted once by the counting variant (same paths, same seed), untimed; its image is the check below
        count_img = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        st = gpu.render_device_stats(params, camera, count_img.data_ptr(), sptr, spp_begin=spp_begin, spp_count=spp_count)
        rays_step = torch.tensor([st["rays"], st["paths"]], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(rays_step)
        total_rays, total_paths = float(rays_step[0].item()), float(rays_step[1].item())

        def step():
            acc.zero_()
            gpu.render_device(params, camera, acc.data_ptr(), sptr, spp_begin=spp_begin, spp_count=spp_count)
            if self.comm is not None:
                self.comm.reduce_to_root(acc.data_ptr(), acc.numel(), 0, sptr)   # RCCL over xGMI: per-GPU sums -> rank 0

        for _ in range(warmup):
            step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        # Frames in flight (--frames, default 2): frame k is ordered on stream k % 2 and accumulates into buffer k % 2, so the
        # last launches of a frame (a few hundred waves of the tail kernel) run beside the first ones of the next; rt_render
        # keeps a workspace per frame in flight.  Every frame is whole inside the timed region (device synchronize below) and
        # is checked against the counting run's image afterwards.  --frames 1: one stream, one buffer, one frame at a time.
        # N > 1: the exchange of frame k runs on a stream of its own, behind that frame's render (the collective is a copy over
        # xGMI, the render is compute: they overlap); every reduce is inside the timed region (barrier + synchronize below)
        n_fl = max(2, args.frames) if (self.comm is not None or args.frames > 1) else 1
        accs = [acc] + [torch.zeros_like(acc) for _ in range(n_fl - 1)]
        streams = [self.stream] + [torch.cuda.Stream(device=dev) for _ in range(n_fl - 1)] if args.frames > 1 else [self.stream] * n_fl
        comm_stream = torch.cuda.Stream(device=dev) if self.comm is not None else None
        rendered = [torch.cuda.Event() for _ in accs]
        reduced = [torch.cuda.Event() for _ in accs]
        extra_warmup = min(warmup, 2) * n_fl if args.frames > 1 else 0
        for k in range(extra_warmup):     # untimed: the second workspace and its streams come into being
            with torch.cuda.stream(streams[k % n_fl]):
                accs[k % n_fl].zero_()
                gpu.render_device(params, camera, accs[k % n_fl].data_ptr(), streams[k % n_fl].cuda_stream, spp_begin=spp_begin, spp_count=spp_count)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            b = k % n_fl
            s_k = streams[b]
            if self.comm is not None and k >= n_fl:
                s_k.wait_event(reduced[b])        # the reduce of the frame that used this buffer last has read it
            with torch.cuda.stream(s_k):
                accs[b].zero_()
            ev[k][0].record(s_k)
            gpu.render_device(params, camera, accs[b].data_ptr(), s_k.cuda_stream, spp_begin=spp_begin, spp_count=spp_count)
            ev[k][1].record(s_k)
            if self.comm is not None:
                rendered[b].record(s_k)
                comm_stream.wait_event(rendered[b])
                self.comm.reduce_to_root(accs[b].data_ptr(), accs[b].numel(), 0, comm_stream.cuda_stream)
                reduced[b].record(comm_stream)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        frame_ms = sum(a.elapsed_time(b) for a, b in ev) / steps           # start to end of a frame on its stream
        # device time per frame: first frame's start to the last end, over the frames (= frame_ms with one frame in flight)
        kernel_ms = max(ev[0][0].elapsed_time(e[1]) for e in ev[-n_fl:]) / steps if n_fl > 1 else frame_ms
        if self.rank != 0:
            return None
        # sanity: the timed kernels' image equals the counting variant's, bit for bit (world 1: nothing was reduced into it)
        if self.world == 1:
            b = count_img.cpu().numpy()
            for a in accs[:min(steps, n_fl)]:
                assert np.array_equal(np.nan_to_num(a.cpu().numpy()), np.nan_to_num(b)), "timed kernel and counting kernel disagree"
        del acc, accs, count_img
        flops = algorithmic_flops(st)
        # HBM-side bytes per step: PMC passes of this very workload (tools/pmc.sh), committed under profiles/
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", PROFILE_ROUND, "%s_%dspp_pmc.json" % (name.lower(), base_spp))
        if self.world == 1 and not width and os.path.exists(prof):
            with open(prof) as f:
                ks = json.load(f)["kernels"].values()
            traffic = sum(2.0 * (k.get("fetch_size_kb") or 0.0) + (k.get("write_size_kb") or 0.0) for k in ks) * 1024.0
            traffic_src = os.path.relpath(prof, ROOT)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        # per kernel: the counting variant's counters split by kernel (rt_last_kernel_stats) x the same constants, over that
        # kernel's busy time per frame in the committed trace.  Kernels of the two pipelines run side by side, so the busy
        # times add up to more than the frame and a kernel's fraction is what it reaches while it shares the chip.
        per_kernel, busy_src = None, None
        if self.world == 1 and not width:
            busy, busy_src = kernel_busy_ms(name, base_spp)
            if busy:
                per_kernel = {}
                for k in KERNELS:
                    f_k = algorithmic_flops(st["kernels"][k])
                    b = busy.get(k)
                    alone = busy.get(k + "_alone")
                    per_kernel[k] = {"flops": f_k, "busy_ms": b, "tflops": (f_k / (b * 1e-3) / 1e12) if b else None,
                                     "frac": (f_k / (b * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS) if b else None,
                                     "alone_ms": alone, "frac_alone": (f_k / (alone * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS) if alone else None}
                per_kernel["fold"] = {"flops": 3.0 * total_paths, "busy_ms": busy.get("fold"), "tflops": None, "frac": None}
        out = {
            "value": total_rays * steps / elapsed / 1e6, "unit": "Mrays/s", "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "frames_rendered": warmup + extra_warmup + steps,      # by the timed kernels (+ one by the counting variant)
            "config": {"workload": "%s %s %dx%d, %d spp per GPU (%d total), depth %d" % (
                name, rt.configs.WORKLOADS[name]["scene"], W, H, spp_count, total_spp, params.max_ray_depth),
                "sharding": ("sample ranges per GPU + RCCL reduce to rank 0, overlapped with the next frame's render" + (", fixed frame" if args.strong else "")) if self.world > 1 else "single GPU",
                "seed": rt.configs.SEED, "primitives": int(flat.desc.n_prims), "bvh_nodes": int(info["n_bvh_nodes"]),
                "bvh": "built on the GPU (%s) in %.2f ms (host flatten %.2f s)" % (args.device_bvh, info["bvh_build_ms"], t_flat) if info["built_on_device"]
                       else "built on the host in %.2f s (flatten)" % t_flat},
            "rays_per_step": total_rays, "paths_per_step": total_paths,
            "node_visits_per_ray": round(st["node_visits"] / max(1, st["rays"]), 2),
            "box_tests_per_ray": round(st["box_tests"] / max(1, st["rays"]), 2),
            "mpaths_per_s": total_paths * steps / elapsed / 1e6,
            # lane occupancy per kernel phase (counting variant): lanes working / lanes issued
            "phase_occupancy": {n: (round(l / s_, 4) if s_ else None) for n, l, s_ in zip(
                ("node_step", "leaf_step", "shade", "path_start", "begin_trace"), st["phase_lanes"], st["phase_slots"])},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                         "kernel": "one rt_render call = the wavefront pipeline (rtk::wf_generate_kernel, wf_traverse_kernel, "
                                   "wf_shade_kernel, wf_tail_kernel, wf_fold_kernel)",
                         "kernel_ms": kernel_ms, "frame_ms": frame_ms, "frames_in_flight": n_fl if args.frames > 1 else 1,
                         "algorithmic_flops_per_launch": flops,
                         "kernels": per_kernel, "kernels_busy_from": busy_src,
                         "note": "no dense contraction -> FP32 vector roofline (SURVEY.md 8d); flops = work counters x per-unit constants; "
                                 "kernel_ms = device time per call: HIP events on the caller's streams from the first call's start to the last one's end, over the calls (the pipelines' streams fork from and join the caller's; with two frames in flight the end of one call runs beside the start of the next, frame_ms = start to end of one call); "
                                 "kernels = per kernel: counters of that kernel (rt_last_kernel_stats) x the same constants over its busy time per frame in the committed kernel trace (busy_ms: the sum of its launches' durations, during which it shares the chip with the other pipeline's kernels) and over the time it needs alone (alone_ms: SQ_BUSY_CYCLES of the PMC passes, whose dispatches run one at a time); "
                                 "traffic = FETCH_SIZE*2 + WRITE_SIZE bytes per step over all kernels, from " + (traffic_src or "n/a")},
        }
        if cpu:
            out.update(self.cpu_baseline(scene, params, camera, sampler, W, H, total_spp, out["value"]))
        return out

    def cpu_baseline(self, scene, params, camera, sampler, W, H, total_spp, gpu_value):
        """the CPU port (oracle/) on all host cores and on one core, median of 3 runs each, on a bounded sample
        of the same workload (whole frame, the first k samples per pixel; k sized to ~6 s per all-core run)."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as O   # cpu_baseline leg only: the oracle is the checker/baseline, never the product
        nproc = os.cpu_count() or 1
        cores = min(16, nproc)   # the GPU box gives one GPU's share of the host: 16 threads
        oracle = O.OracleScene(scene)

        def timed(spp_count, threads, tile=None):
            c0 = time.perf_counter()
            _, rays = oracle.render(params, camera, sampler, spp_begin=0, spp_count=spp_count, threads=threads, **({"tile": tile} if tile else {}))
            dt = time.perf_counter() - c0
            return rays / dt / 1e6, dt

        _, cal = timed(1, cores)                                     # calibration, untimed
        k_all = max(1, min(total_spp, int(6.0 / max(cal, 1e-3))))
        runs_all = [timed(k_all, cores) for _ in range(3)]
        # one core: the same frame's top rows (a 1/8 strip), 1 sample or more, ~4 s per run
        strip = (0, 0, W, max(8, H // 8))
        _, cal1 = timed(1, 1, strip)
        k_one = max(1, min(total_spp, int(4.0 / max(cal1, 1e-3))))
        runs_one = [timed(k_one, 1, strip) for _ in range(3)]
        v_all, v_one = statistics.median(r[0] for r in runs_all), statistics.median(r[0] for r in runs_one)
        return {
            "cpu_baseline": {"value": v_all, "unit": "Mrays/s", "cores": cores, "kind": "port", "nproc": nproc,
                             "runs": [round(r[0], 3) for r in runs_all], "one_core": {"value": v_one, "runs": [round(r[0], 3) for r in runs_one],
                                      "sample": "rows 0..%d of the frame, samples 0..%d, 1 thread, %.1f s per run" % (strip[3], k_one, runs_one[0][1])},
                             "sample": "%dx%d whole frame, samples 0..%d of %d, depth %d, %d threads, median of 3 runs of %.1f s" % (
                                 W, H, k_all, total_spp, params.max_ray_depth, cores, runs_all[0][1])},
            "speedup_vs_cpu_port": gpu_value / v_all,
        }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C1")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--frames", type=int, default=2, choices=[1, 2, 3], help="frames in flight: 2 = consecutive frames on two streams and two buffers (default), 1 = one at a time (3: A/B only, the library keeps two workspaces)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="skip the C2 / C3 / C4 sub-workloads of the default run")
    ap.add_argument("--sub", default=None, help="comma-separated sub-workloads instead of C2,C3,C4")
    ap.add_argument("--strong", action="store_true", help="N>1: split the workload's spp over the ranks (fixed frame) instead of N x spp")
    ap.add_argument("--device-bvh", nargs="?", const="ploc", default=None, choices=["ploc", "lbvh"],
                    help="build the BVH on the GPU (rt_bvh_build.hip) instead of on the host: clustering (default) or Morton radix tree")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if world == 0:
        if args.gpus > 1:
            spawn_ranks(args.gpus)      # never returns: no GPU call has happened in this process
        world = 1
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report one as the other" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.frames > 1:
        # two frames in flight = two caller streams + two pipeline streams per frame: with HIP's default of 4 hardware queues
        # per process they share queues and run one after the other (profiles/r3/ab_frames_in_flight.txt); read at HIP start-up
        # (the GPU boxes export the default, 4, so the variable is raised rather than defaulted)
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 4 * args.frames:
            os.environ["GPU_MAX_HW_QUEUES"] = str(4 * args.frames)
    import torch
    import rtiow_loader
    rt = rtiow_loader.load()
    dist = None
    if world > 1 or os.environ.get("RT_BENCH_FORCE_DIST"):   # the env switch rehearses the RCCL calls with one rank on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    n_gpus = dist.get_world_size() if dist is not None else 1      # reported from the initialised group, never from the flag

    B = Bench(args, rt, torch, dist, rank, local_rank, world)
    head = B.run(args.workload, args.steps, args.warmup, spp=args.spp, width=args.width, cpu=not args.no_cpu and world == 1)
    subs = {}
    default_run = args.workload == "C1" and not args.spp and not args.width
    if world == 1 and not args.no_sub and (default_run or args.sub):
        names = [s for s in (args.sub.split(",") if args.sub else [w[0] for w in SUB_WORKLOADS]) if s]
        table = {w[0]: w for w in SUB_WORKLOADS}
        for nme in names:
            _, k, wu = table.get(nme, (nme, 3, 1))
            r = B.run(nme, k, wu)
            subs[nme] = {"mrays_per_s": r["value"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "workload": r["config"]["workload"],
                         "rays_per_step": r["rays_per_step"], "node_visits_per_ray": r["node_visits_per_ray"],
                         "box_tests_per_ray": r["box_tests_per_ray"], "roofline_frac": r["roofline"]["frac"],
                         "achieved_tflops": r["roofline"]["achieved"], "kernel_ms": r["roofline"]["kernel_ms"], "traffic": r["roofline"]["traffic"],
                         "kernels": r["roofline"]["kernels"],
                         "bvh": r["config"]["bvh"], "bvh_nodes": r["config"]["bvh_nodes"], "primitives": r["config"]["primitives"],
                         "phase_occupancy": r["phase_occupancy"]}
    if rank == 0:
        out = {"metric": "Mrays/sec", "value": head["value"], "unit": "Mrays/s", "n_gpus": n_gpus, "steps": head["steps"], "warmup": head["warmup"],
               "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        out.update({k: v for k, v in head.items() if k not in out})
        if subs:
            out["workloads"] = subs
        print(json.dumps(out))
    if B.comm is not None:
        B.comm.close()
    if dist is not None:
        dist.destroy_process_group()
    B.gpu.close()


if __name__ == "__main__":
    main()
