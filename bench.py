#!/usr/bin/env python3
"""Headline benchmark: camera-path samples/s of the path-tracing hot path on MI355X.

Workload (the configuration the headline metric is quoted on):
scene/buddha at 1920x1080, 256 samples per pixel, depth 12, clamp 10, aperture 0.1, the scene's
three area lights + distant light + environment map with MIS, a ~1.09 M-triangle mesh with the
microfacet plastic material (Kd .6 / Ks 1 / Pr 0.27 / Ni 1.5).  The Happy Buddha OBJ is not part of
the reference snapshot, so a generated stand-in of the same triangle count is used unless
scenes/buddha/happy-buddha.obj exists.  Synthetic data, no files are read from the reference.

One step = one pass of the hot path over the whole frame: rrt_render of 256 samples of every pixel
(primary, then start / extend / shade / connect per iteration, then film accumulation), scene and film
resident in HBM before the clock starts.  With N GPUs (launched by torch.distributed.run, one rank per GPU)
rank r renders sample indices [256 r, 256 (r+1)) and the per-GPU films are summed onto rank 0 with
one RCCL reduce inside the timed region: weak scaling, value = all ranks' camera paths / max time.

    python bench.py --gpus 1 --steps 3 --warmup 1
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _load(name, rel):
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, rel)
    kw = {"submodule_search_locations": [os.path.dirname(path)]} if rel.endswith("__init__.py") else {}
    spec = importlib.util.spec_from_file_location(name, path, **kw)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


rr = _load("rust_raytracer_amd", "rust-raytracer_amd/__init__.py")
standin = _load("rust_raytracer_amd.standin", "rust-raytracer_amd/standin.py")
rrdist = _load("rust_raytracer_amd.distributed", "rust-raytracer_amd/distributed.py")
stress = _load("rust_raytracer_amd.stress", "rust-raytracer_amd/stress.py")
rrbuild = _load("rust_raytracer_amd.build", "rust-raytracer_amd/build.py")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# The rate at which one MI355X fetches 128-byte lines that lanes pick at random (tools/fetch_probe.hip: 40-44 G lines/s from
# a 4 GiB table and from one resident in the Infinity Cache alike): what scattered traffic runs against, rather than bytes.
SCATTERED_LINES_PEAK_G = 42.0
# algorithmic bytes per ray: 64 B per BVH node visited (two boxes of six f32 and the links, the figure the roofline is
# defined on), 36 B per triangle tested (nine f32 positions), 32 B ray in, 16 B hit out.  The node as stored is smaller
# (48 B: four links, four boxes of six bytes, corner and grid shifts); the roofline entry carries the rate on that as well.
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 64, 36, 32, 16
NODE_BYTES_STORED, TRI_BYTES_STORED = 48, 48
STATE_BYTES_PER_BOUNCE = 264  # wavefront path state read + written per shaded bounce


def load_buddha(width, height, hard=False):
    folder = os.path.join(rr.SCENES, "buddha")
    rs = rr.load_render_settings(folder)
    # the "microfacet" configuration: Plastic -> Lambertian + Trowbridge-Reitz with alpha = 0.27^2
    material = rr.make_material(rr.MAT_PLASTIC, kd=(0.6, 0.6, 0.6), ks=(1.0, 1.0, 1.0), roughness=0.27, ior=1.5)
    if os.path.exists(os.path.join(folder, "happy-buddha.obj")) and not hard:
        scene = rr.Scene.load_from_folder(folder)
        scene.set_material(0, material)
        mesh = "happy-buddha.obj"
    elif hard:
        pos, nrm = standin.happy_buddha_hard()
        scene = rr.Scene.load_from_folder_with_mesh(folder, pos, nrm, material)
        mesh = "generated stand-in with overhanging tiers, through-holes, hashed triangle order"
    else:
        # RRT_BENCH_MESH_DIVISOR: diagnostic only (coarser stand-in = smaller working set); the bench line is quoted at 1
        div = int(os.environ.get("RRT_BENCH_MESH_DIVISOR", "1"))
        pos, nrm = standin.happy_buddha_standin() if div == 1 else standin.happy_buddha_standin(966 // div, 563 // div)
        scene = rr.Scene.load_from_folder_with_mesh(folder, pos, nrm, material)
        mesh = "generated stand-in"
    camera = rr.Camera.from_settings(rs, width, height)
    return scene, rs, camera, mesh


def make_workload(name, W, H, SPP, spp_label, trios, rank=0):
    """(scene, render settings or None, camera, mesh label, depth limit, clamp, description) of a named workload."""
    if name == "stress":
        t_gen = time.perf_counter()
        scene, cam_args, depth_limit, clamp = stress.stress_scene(rr, trios)
        camera = rr.Camera(image_width=W, image_height=H, **cam_args)
        if rank == 0:
            print("stress scene generated in %.1f s" % (time.perf_counter() - t_gen), file=sys.stderr, flush=True)
        return (scene, None, camera, "%d generated trios" % trios, depth_limit, clamp,
                "stress: %d cow/lamp/teapot trios on a grid, %dx%d, %d spp %s, depth %d, clamp %g, matte, distant + area light" % (
                    trios, W, H, SPP, spp_label, depth_limit, clamp))
    if name.startswith("cornell"):
        # scene/cornell at the frame size and sample count of the second configuration (--spp 1024); "-matte": every
        # material replaced by Matte with its Kd (the "matte + area light" reading), else the materials as shipped
        folder = os.path.join(rr.SCENES, "cornell")
        rs = rr.load_render_settings(folder)
        scene = rr.Scene.load_from_folder(folder)
        if name == "cornell-matte":
            d = scene.desc()
            for i in range(d.n_materials):
                m = d.materials[i]
                scene.set_material(i, rr.make_material(rr.MAT_MATTE, kd=tuple(m.kd), roughness=0.0))
        camera = rr.Camera.from_settings(rs, W, H)
        return (scene, rs, camera, "cornell.obj", rs.depth_limit, rs.clamp,
                "scene/cornell %dx%d, %d spp %s, depth %d, clamp %g, %s, area light" % (
                    W, H, SPP, spp_label, rs.depth_limit, rs.clamp,
                    "all materials matte" if name == "cornell-matte" else "materials as shipped"))
    scene, rs, camera, mesh = load_buddha(W, H, hard=name == "buddha-hard")
    return (scene, rs, camera, mesh, rs.depth_limit, rs.clamp,
            "scene/buddha %dx%d, %d spp %s, depth %d, clamp %g, aperture %g, microfacet plastic, "
            "3 area + distant + environment lights with MIS" % (W, H, SPP, spp_label, rs.depth_limit, rs.clamp, rs.aperture))


def host_cpus():
    """(logical CPUs of the machine, CPUs this process may use).  On a shared box the second is smaller: the
    cgroup's quota (cpu.max), else the thread count the box exports for worker pools (OMP_NUM_THREADS), else the
    scheduler affinity mask."""
    total = os.cpu_count() or 1
    usable = total
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = min(usable, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        usable = min(usable, max(1, int(os.environ.get("OMP_NUM_THREADS", usable))))
    except ValueError:
        pass
    return total, usable


def cpu_baseline(scene, rs, width, height, threads, budget_paths, budget_seconds=0.0):
    """Times the CPU oracle (port of the reference algorithm) on a bounded sample of the same workload: the same
    scene, camera, frame size, depth and clamp, handed out in the reference's 128 x 128 buckets (src/film.rs:134-148;
    135 of them at 1920 x 1080, so every worker thread has buckets to take), at fewer samples per pixel."""
    from oracle import binding as ob
    w, h = width, height
    cam = rr.Camera.from_settings(rs, w, h)
    orc = ob.Oracle(scene.desc())
    orc.set_camera(cam)
    s = rr.make_settings(w, h, rs.depth_limit, rs.clamp, seed=0)
    bucket = (128, 128)
    if budget_seconds > 0:
        # pilot pass to size the sample for about budget_seconds of CPU work
        _, pilot = orc.render(s, 0, 1, threads=threads, bucket=bucket)
        spp = int(min(4096, max(1, budget_seconds * pilot.paths / max(pilot.seconds, 1e-3) / (w * h))))
    else:
        spp = max(1, budget_paths // (w * h))
    _, st = orc.render(s, 1, 1 + spp, threads=threads, bucket=bucket)
    rays = st.rays_extend + st.rays_shadow
    return {
        "value": st.paths / st.seconds,
        "unit": "samples/s",
        "cores": int(st.threads_busy),          # worker threads that rendered at least one bucket
        "threads_started": int(st.threads),
        "host_cores": host_cpus()[0],           # logical CPUs of the machine
        "host_cores_usable": host_cpus()[1],    # of which this process may use (quota / affinity / the box's pool size)
        "buckets": int(st.buckets),
        "kind": "port",
        "sample": "same scene/camera/depth at %dx%d in %d buckets of %dx%d, %d spp (%d camera paths), %.1f s" % (
            w, h, st.buckets, bucket[0], bucket[1], spp, st.paths, st.seconds),
        "mrays_per_s": rays / st.seconds * 1e-6,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: seconds of oracle work (0 = use --cpu-paths)")
    ap.add_argument("--cpu-paths", type=int, default=0, help="CPU baseline: camera paths when --cpu-seconds is 0 (both 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline worker threads (0 = every CPU this process may use)")
    ap.add_argument("--batch-samples", type=int, default=0)
    ap.add_argument("--max-slots", type=int, default=0, help="paths in flight (0 = library default: sized from free device memory)")
    ap.add_argument("--no-calibrate", action="store_true")
    ap.add_argument("--workload", choices=["buddha", "buddha-hard", "stress", "cornell", "cornell-matte"], default="buddha",
                    help="buddha: the headline configuration; buddha-hard: the same with the stand-in mesh that has overhanging "
                         "tiers, through-holes and hashed triangle order; stress: cow/lamp/teapot trios on a grid (depth 16)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the extra lines the default run adds under other_workloads (cornell 1080p / 1024 spp, buddha-hard)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --spp samples per pixel on EVERY GPU (rank r renders [r spp, (r+1) spp)); strong: --spp in total, "
                         "rank r renders [r spp / N, (r+1) spp / N) (the 3840x2160, 4096 spp configuration: --scaling strong "
                         "--width 3840 --height 2160 --spp 4096)")
    ap.add_argument("--trios", type=int, default=3200, help="stress workload: number of trios (15,632 triangles each)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" %
                             (args.gpus, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    W, H, SPP = args.width, args.height, args.spp
    spp_label = "per GPU" if args.scaling == "weak" else "in total over all GPUs"
    scene, rs, camera, mesh, depth_limit, clamp, workload = make_workload(args.workload, W, H, SPP, spp_label, args.trios, rank)
    ctx = rr.Context(local_rank)
    if os.environ.get("RRT_DEVICE_BVH_OVERRIDE"):  # diagnostic: 0 host build, 1 device build (default: by triangle count)
        ctx.set_option(rr.OPT_DEVICE_BVH, int(os.environ["RRT_DEVICE_BVH_OVERRIDE"]))
    t_up = time.perf_counter()
    ctx.upload_scene(scene)
    upload_s = time.perf_counter() - t_up
    ctx.set_camera(camera)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
    ctx.film_bind(film.data_ptr(), W, H)
    stream = torch.cuda.current_stream(device)
    ctx.set_stream(ctypes.c_void_p(stream.cuda_stream))
    if os.environ.get("RRT_TAIL_PATHS_OVERRIDE"):  # diagnostic: threshold of the tail kernel
        ctx.set_option(rr.OPT_TAIL_PATHS, int(os.environ["RRT_TAIL_PATHS_OVERRIDE"]))
    if args.max_slots:
        ctx.set_option(rr.OPT_MAX_SLOTS, args.max_slots)
    if args.batch_samples:
        ctx.set_option(rr.OPT_BATCH_SAMPLES, args.batch_samples)
    settings = rr.make_settings(W, H, depth_limit, clamp, seed=0)
    if args.scaling == "strong":
        s_begin, s_end = rrdist.sample_range(rank, world, SPP)
    else:
        s_begin, s_end = rrdist.weak_sample_range(rank, SPP)
    total_spp = SPP if args.scaling == "strong" else SPP * world

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def calibrate(settings, s_begin, s_end):
        """Untimed: nodes / triangles fetched per ray of each traversal kernel and the share of rays that go into the
        hierarchy at all, for the roofline's algorithmic bytes (a counting build of the same kernels; per-ray means
        do not depend on spp)."""
        ctx.set_option(rr.OPT_COUNT_TRAVERSAL, 1)
        ctx.reset_stats()
        film.zero_()
        ctx.render(settings, s_begin, s_begin + min(8, s_end - s_begin))
        st = ctx.stats()
        rays = (st.rays_primary, st.rays_extend - st.rays_primary, st.rays_shadow)
        per_ray = [(st.nodes_by_kernel[k] / max(rays[k], 1), st.tris_by_kernel[k] / max(rays[k], 1),
                    st.rays_entered_by_kernel[k] / max(rays[k], 1)) for k in range(3)]
        ctx.set_option(rr.OPT_COUNT_TRAVERSAL, 0)
        return per_ray

    def timed(settings, s_begin, s_end, steps, warmup, reduce=True):
        """`warmup` untimed steps, then `steps` timed ones between fences; returns (seconds, stats of the timed steps)."""
        def step():
            film.zero_()
            ctx.render(settings, s_begin, s_end)
            if reduce:
                rrdist.reduce_film(film, dst=0)  # RCCL sum onto rank 0; a no-op on one GPU
        for _ in range(warmup):
            step()
        fence()
        ctx.set_option(rr.OPT_TIMING, 1)
        ctx.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        seconds = time.perf_counter() - t0
        st = ctx.stats()
        ctx.set_option(rr.OPT_TIMING, 0)
        return seconds, st

    per_ray = calibrate(settings, s_begin, s_end) if rank == 0 and not args.no_calibrate else None
    elapsed, st = timed(settings, s_begin, s_end, args.steps, args.warmup)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # untimed: one more frame with every kernel on the one stream, for the dominant kernel's launch time when it
    # has the GPU to itself (in the timed frames it shares it with the previous iteration's visibility rays)
    alone_ms = None
    if rank == 0 and not args.no_calibrate:
        keep = film.clone()
        ctx.set_option(rr.OPT_SIDE_STREAM, 0)
        ctx.set_option(rr.OPT_TIMING, 1)
        ctx.reset_stats()
        ctx.render(settings, s_begin, s_end)
        torch.cuda.synchronize(device)
        st1 = ctx.stats()
        alone_ms = st1.ms_extend / max(st1.launches_extend, 1)
        ctx.set_option(rr.OPT_SIDE_STREAM, 1)
        ctx.set_option(rr.OPT_TIMING, 0)
        film.copy_(keep)

    paths_all = float(W) * H * total_spp * args.steps
    value = paths_all / elapsed
    rays_rank = st.rays_extend + st.rays_shadow
    out = None
    if rank == 0:
        # sanity: the film holds `world` x SPP samples per pixel and finite radiance
        fsum = film.sum(dim=(0, 1)).tolist()
        assert abs(fsum[3] - float(W) * H * total_spp) < 1e-3 * W * H * total_spp, fsum
        assert all(x == x and x > 0 for x in fsum[:3]), fsum
        roofline = None
        roofline_shade = None
        kernels = None
        rays_traversed = None
        default_workload = args.workload == "buddha" and (args.width, args.height, args.spp) == (1920, 1080, 256)
        # measured with the PMC counters on the default workload (tools/profile_bench.sh, rocprofv3 --pmc runs of their
        # own: the counters cannot be read inside this run): HBM bytes and L2 hit rate per kernel over one frame; no
        # figure for other workloads -- and none when the kernel sources have changed since that profile was taken
        prof, prof_note = None, None
        tpath = os.path.join(ROOT, "profiles", "extend_traffic.json")
        if default_workload and os.path.exists(tpath):
            with open(tpath) as f:
                prof = json.load(f)
            if prof.get("kernel_fingerprint") != rrbuild.kernel_fingerprint():
                prof_note = "profiles/extend_traffic.json (%s) was taken from other kernel sources than this tree's" % prof.get("source")
                prof = None
            else:
                prof_note = "PMC counters of rocprofv3 run %s of this command, same kernel sources" % prof.get("source")
        if per_ray is not None and st.launches_extend:
            def entry(name, k, n_rays, launches, ms):
                bytes_per_ray = per_ray[k][0] * NODE_BYTES + per_ray[k][1] * TRI_BYTES + RAY_BYTES + HIT_BYTES
                gbs = n_rays * bytes_per_ray / (ms * 1e-3) * 1e-9 if ms > 0 else 0.0
                fetched_per_ray = per_ray[k][0] * NODE_BYTES_STORED + per_ray[k][1] * TRI_BYTES_STORED + RAY_BYTES + HIT_BYTES
                return {"kernel": name, "bytes_fetched_per_ray": fetched_per_ray,
                        "fetched_gb_per_s": n_rays * fetched_per_ray / (ms * 1e-3) * 1e-9 if ms > 0 else 0.0, "rays_per_launch": n_rays / max(launches, 1), "bytes_per_ray": bytes_per_ray,
                        "nodes_per_ray": per_ray[k][0], "tris_per_ray": per_ray[k][1], "share_of_rays_entering_the_hierarchy": per_ray[k][2],
                        "launches": int(launches), "avg_launch_ms": ms / max(launches, 1), "algorithmic_gb_per_s": gbs,
                        "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
            kernels = [entry("k_primary", 0, st.rays_primary, st.launches_primary, st.ms_primary),
                       entry("k_extend", 1, st.rays_extend - st.rays_primary, st.launches_extend, st.ms_extend),
                       entry("k_connect", 2, st.rays_shadow, st.launches_extend + st.launches_primary, st.ms_connect)]
            # rays that did more than fail the test against the box around the model
            rays_traversed = (st.rays_primary * per_ray[0][2] + (st.rays_extend - st.rays_primary) * per_ray[1][2] +
                              st.rays_shadow * per_ray[2][2])
            # the dominant kernel by time: k_extend (continuation rays)
            e = kernels[1]
            traffic = l2 = hbm_gbs = None
            if prof is not None and prof.get("hbm_bytes_per_frame"):
                traffic = prof["hbm_bytes_per_frame"] * args.steps / e["launches"]
                l2 = prof.get("l2_hit_rate")
                hbm_gbs = traffic / (e["avg_launch_ms"] * 1e-3) * 1e-9
            roofline = {"kernel": "k_extend", "bound": "hbm", "achieved": e["algorithmic_gb_per_s"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": e["algorithmic_gb_per_s"] / HBM_PEAK_GBS, "traffic": traffic,
                        "hbm_gb_per_s": hbm_gbs, "l2_hit_rate": l2,
                        "bytes_per_ray": e["bytes_per_ray"], "nodes_per_ray": e["nodes_per_ray"],
                        # the same with the records as stored (48-byte nodes, 48-byte triangle records)
                        "bytes_fetched_per_ray": e["bytes_fetched_per_ray"], "fetched_gb_per_s": e["fetched_gb_per_s"],
                        "frac_fetched": e["fetched_gb_per_s"] / HBM_PEAK_GBS,
                        "tris_per_ray": e["tris_per_ray"], "launches": e["launches"], "avg_launch_ms": e["avg_launch_ms"],
                        "rays_per_launch": e["rays_per_launch"]}
            k = (prof or {}).get("kernels", {}).get("k_extend")
            if k and k.get("read_lines_per_frame") is not None:
                # the same traffic as memory-side requests of the L2s (one per 128-byte line read, one per 64 bytes
                # written) over the launches' own time, against the probe's rate for scattered lines
                reqs = (k["read_lines_per_frame"] + k["write_requests_per_frame"]) * args.steps / e["launches"]
                rate = reqs / (e["avg_launch_ms"] * 1e-3) * 1e-9
                roofline["memory_requests"] = {"per_launch": reqs, "g_per_s": rate, "peak_g_per_s": SCATTERED_LINES_PEAK_G,
                                               "frac": rate / SCATTERED_LINES_PEAK_G}
            if prof_note:
                roofline["traffic_source"] = prof_note
            if alone_ms:
                roofline["avg_launch_ms_alone"] = alone_ms
                roofline["frac_alone"] = roofline["frac"] * e["avg_launch_ms"] / alone_ms
                if "memory_requests" in roofline:
                    roofline["memory_requests"]["frac_alone"] = roofline["memory_requests"]["frac"] * e["avg_launch_ms"] / alone_ms
        if st.ms_shade > 0:
            # shade streams path state: per shaded bounce 2 x (ray 32 + throughput 12 + radiance 12 + pixel / flags 8 +
            # rng 8 + hit 16 + shadow ray 44) = 264 algorithmic bytes
            gbs = st.path_bounces * STATE_BYTES_PER_BOUNCE / (st.ms_shade * 1e-3) * 1e-9
            roofline_shade = {"kernel": "k_shade", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": gbs / HBM_PEAK_GBS, "bytes_per_bounce": STATE_BYTES_PER_BOUNCE,
                              "bounces_per_step": st.path_bounces / args.steps, "ms_per_step": st.ms_shade / args.steps,
                              "traffic": None, "hbm_gb_per_s": None, "l2_hit_rate": None}
            k = (prof or {}).get("kernels", {}).get("k_shade")
            if k:
                roofline_shade["traffic"] = k["hbm_bytes_per_frame"]   # per step (one frame), all its launches
                roofline_shade["hbm_gb_per_s"] = k["hbm_bytes_per_frame"] / (st.ms_shade / args.steps * 1e-3) * 1e-9
                roofline_shade["l2_hit_rate"] = k.get("l2_hit_rate")
                if k.get("read_lines_per_frame") is not None:
                    reqs = k["read_lines_per_frame"] + k["write_requests_per_frame"]
                    rate = reqs / (st.ms_shade / args.steps * 1e-3) * 1e-9
                    roofline_shade["memory_requests"] = {"per_step": reqs, "g_per_s": rate, "peak_g_per_s": SCATTERED_LINES_PEAK_G,
                                                         "frac": rate / SCATTERED_LINES_PEAK_G}
        out = {
            "metric": "camera-path samples/sec",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "mesh": mesh, "triangles": int(scene.desc().n_tris),
                       "bvh_nodes": int(st.bvh_nodes), "scene_bytes": int(st.bvh_bytes), "scene_upload_s": upload_s,
                       "sharding": "sample index ranges per GPU", "samples_per_pixel_total": total_spp,
                       "samples_per_pixel_this_rank": s_end - s_begin},
            "mrays_per_s": rays_rank * world / elapsed * 1e-6,
            # without the camera rays that only failed the test against the box around the model
            "mrays_per_s_traversed": rays_traversed * world / elapsed * 1e-6 if rays_traversed is not None else None,
            "rays_per_path": rays_rank / float(st.paths) if st.paths else None,
            "bounces_per_path": st.path_bounces / float(st.paths) if st.paths else None,
            "kernel_ms_per_step": {"primary": st.ms_primary / args.steps, "start": st.ms_raygen / args.steps,
                                   "extend": st.ms_extend / args.steps,
                                   "shade": st.ms_shade / args.steps, "connect": st.ms_connect / args.steps,
                                   "film": st.ms_film / args.steps, "tail": st.ms_tail / args.steps},
            "roofline": roofline,
            "roofline_shade": roofline_shade,
            "traversal_kernels": kernels,
        }
        if world == 1 and rs is not None and (args.cpu_paths > 0 or args.cpu_seconds > 0):
            threads = args.cpu_threads or host_cpus()[1]
            out["cpu_baseline"] = cpu_baseline(scene, rs, W, H, threads, args.cpu_paths, args.cpu_seconds)
            out["cpu_baseline"]["all_threads_busy"] = out["cpu_baseline"]["cores"] == min(threads, out["cpu_baseline"]["buckets"])
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        # ---- the other single-GPU configurations, on the same context: a few frames each, untimed parts left out
        if world == 1 and default_workload and not args.no_other_workloads and not args.no_calibrate:
            others = {}
            for key, name, spp_o, steps_o in (("cornell_1080p_1024spp", "cornell", 1024, 2), ("buddha_hard_standin_1080p_256spp", "buddha-hard", 256, 5)):
                scene_o, rs_o, camera_o, mesh_o, depth_o, clamp_o, workload_o = make_workload(name, W, H, spp_o, "per GPU", args.trios)
                ctx.upload_scene(scene_o)
                ctx.set_camera(camera_o)
                settings_o = rr.make_settings(W, H, depth_o, clamp_o, seed=0)
                seconds, st_o = timed(settings_o, 0, spp_o, steps_o, 1, reduce=False)
                rays_o = st_o.rays_extend + st_o.rays_shadow
                others[key] = {"workload": workload_o, "mesh": mesh_o, "triangles": int(scene_o.desc().n_tris), "steps": steps_o, "warmup": 1,
                               "value": float(W) * H * spp_o * steps_o / seconds, "unit": "samples/s",
                               "ms_per_step": seconds / steps_o * 1e3, "mrays_per_s": rays_o / seconds * 1e-6,
                               "rays_per_path": rays_o / float(st_o.paths), "bounces_per_path": st_o.path_bounces / float(st_o.paths),
                               "kernel_ms_per_step": {"primary": st_o.ms_primary / steps_o, "start": st_o.ms_raygen / steps_o,
                                                      "extend": st_o.ms_extend / steps_o, "shade": st_o.ms_shade / steps_o,
                                                      "connect": st_o.ms_connect / steps_o, "film": st_o.ms_film / steps_o,
                                                      "tail": st_o.ms_tail / steps_o}}
            out["other_workloads"] = others
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
