#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mrays/s (and fps) of whole-frame rendering at 1920x1080,
primary + shadow rays, on the 100k-triangle BIH scene (BASELINE.json configs[2] = SURVEY.md scene S3).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one frame: every pixel's primary ray + its shadow rays through the hot path (raygen -> BIH closest hit ->
shadow any-hit -> shade), scene resident in HBM.  With N GPUs the frame's 65x65 reference tiles are sharded round-robin
and gathered to rank 0 with one RCCL gather (strong scaling: the frame is fixed).  Rank 0 prints ONE JSON line.

The line carries `roofline` (dominant kernel vs the 8 TB/s HBM roofline, algorithmic bytes per SURVEY.md 8(d), kernel
time from HIP events recorded on the launch stream over the timed region) and `cpu_baseline` (the CPU oracle -- a C++
restatement of the reference algorithm, kind "port" -- timed on this box's host cores on a bounded tile sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="S3", help="S1 S2 S3 S3mesh S4 S5 (default: the headline workload S3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--mode", type=int, default=0, help="0 = renderTile (1 ray/pixel), 1 = renderTileSubsample (adaptive)")
    ap.add_argument("--lanes", type=int, default=4, help="launches kept in flight per GPU (HIP streams / context slots)")
    ap.add_argument("--time-every", type=int, default=1, help="HIP-event pair on every k-th launch of the timed region (roofline kernel time)")
    ap.add_argument("--host-build", action="store_true", help="build the BIH with the host builder (glome_sb_bih) instead of on the GPU")
    ap.add_argument("--force-dist", action="store_true", help="one GPU, but through the multi-GPU pipeline with a one-rank RCCL group (rehearsal)")
    ap.add_argument("--group", type=int, default=0, help="frames per launch (and per RCCL gather); default by rank count and run length (up to 4 on one or two GPUs, 8 on more)")
    ap.add_argument("--product", default="packed", choices=["packed", "rgbad"],
                    help="what a frame is: GlomeView's framebuffer of packed 0x00RRGGBB pixels (blitTile; 4 B/pixel cross xGMI) "
                         "or the float (r,g,b,a,depth) tuples (20 B/pixel)")
    args = ap.parse_args()

    # the frames in flight run on separate HIP streams; give them separate hardware queues (the runtime's default is 4,
    # which makes the frame period depend on how the streams happen to share queues)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch  # device memory, streams, torch.distributed (RCCL)

    from glome_amd import _lib as L
    from glome_amd import api, dist, scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU path to time)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.force_dist and world == 1:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        tdist.init_process_group("nccl", device_id=device)

    cfg = scenes.CONFIGS[args.scene]
    sd = cfg["make"]()
    W, H, maxdepth = cfg["width"], cfg["height"], cfg["maxdepth"]
    t0 = time.time()
    b = api.Builder()
    ctx = api.Context(local)

    class _DeviceBih:  # the builder with the trees of large `bih` / `mesh` calls made on the GPU (same trees)
        def __getattr__(self, name):
            return getattr(b, name)

        def bih(self, ids):
            return ctx.bih(b, ids)[0] if len(ids) >= 4096 and not args.host_build else b.bih(ids)

        def mesh(self, verts, norms, tris, mats):
            return ctx.mesh(b, verts, norms, tris, mats)[0] if len(tris) >= 4096 and not args.host_build else b.mesh(verts, norms, tris, mats)
    nmap, _ = sd.replay(_DeviceBih())
    scene = ctx.commit(b, nmap[sd.root])
    setup_s = time.time() - t0
    info = scene.info()
    cam = api.camera(*sd.cam)
    lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
    P = api.render_params(width=W, height=H, maxdepth=maxdepth, mode=args.mode)

    if args.group <= 0:
        # frames per launch (and per gather).  Deep batches pay a fill / drain of about one launch per run, so short runs get
        # shallower ones; the break-even points are measured (tools/short_run_groups.py, profiles/r01_l_short_run_groups.log):
        # at 8 ranks a 20-step run takes 0.091 ms per step with one frame per launch and 0.052 with four.
        if args.mode != 0 or args.product != "packed":
            args.group = 1
        elif world == 1:
            args.group = max(1, min(4, args.steps // 24))
        elif world == 2:
            args.group = 1 if args.steps < 8 else (4 if args.steps < 64 else 8)
        elif world <= 4:
            args.group = 2 if args.steps < 16 else 8
        else:
            args.group = 4 if args.steps < 40 else 8
    sf = dist.ShardedFrame(scene, P, rank, world, device, lanes=args.lanes, product=args.product, group=args.group, force_pipeline=args.force_dist)

    def barrier():
        if dist_on:
            tdist.barrier()
        torch.cuda.synchronize(device)

    # ray count of one frame (identical every step: the scene and camera are fixed)
    st = sf.step(cam, lights, stats=True)
    rays_local = torch.tensor([st["rays_primary"], st["rays_shadow"], st["rays_secondary"]], dtype=torch.float64, device=device) if dist_on else None
    if dist_on:
        tdist.all_reduce(rays_local)
        rays = [int(x) for x in rays_local.tolist()]
    else:
        rays = [st["rays_primary"], st["rays_shadow"], st["rays_secondary"]]
    rays_per_step = sum(rays)

    sf.prime(cam, lights)  # every lane's slot / stream / kernel instance exists before the warm-up (initialisation, not steps)
    for _ in range(args.warmup):
        sf.step(cam, lights)
    sf.flush()
    barrier()
    # every --time-every-th launch of the timed region carries a HIP-event pair on its launch stream (default: every launch)
    ctx.lib.glome_ctx_timing_begin_sampled(ctx.h, args.steps, max(1, args.time_every))
    t_start = time.perf_counter()
    for _ in range(args.steps):
        sf.step(cam, lights)
    sf.flush()  # frames are pipelined (several in flight): complete the last ones inside the timed region
    barrier()
    elapsed = time.perf_counter() - t_start
    kms = np.zeros(args.steps, np.float32)
    nk = ctx.lib.glome_ctx_timing_end(ctx.h, kms.ctypes.data_as(L.c_fp), args.steps)
    if dist_on:
        e = torch.tensor([elapsed], dtype=torch.float64, device=device)
        tdist.all_reduce(e, op=tdist.ReduceOp.MAX)
        elapsed = float(e.item())

    if rank != 0:
        if dist_on:
            tdist.barrier()
            tdist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = rays_per_step / (elapsed / args.steps) / 1e6

    # ---- roofline of the dominant kernel (rank 0's render launch) ----
    # algorithmic bytes: 32 B ray in + 32 B hit out per closest-hit ray, 32 + 4 per shadow ray,
    # 16 B per BIH node visited, S_prim per primitive tested (48 B triangle, 16 B sphere); node / primitive visits are
    # the reference algorithm's (no early-out, rayint_debug convention, Bih.hs:378-412), counted by the faithful kernel
    # on this very frame (tests pin those counts to the CPU oracle's).
    Pl = dist._clone_params(P, tile_first=(rank if world > 1 else 0), tile_stride=world)
    ctx.lib.glome_ctx_use_slot(ctx.h, None, 0)
    tmp = torch.zeros((H, W, 5), dtype=torch.float32, device=device)
    torch.cuda.synchronize(device)
    Pf = dist._clone_params(Pl, faithful=1, count_work=1)
    stf = scene.render_dev(cam, lights, Pf, tmp.data_ptr())
    Pc = dist._clone_params(Pl, faithful=0, count_work=1)
    stc = scene.render_dev(cam, lights, Pc, tmp.data_ptr())
    s_prim = 48 if info["n_triangles"] >= info["n_spheres"] else 16
    node_b = 64 if info["n_mesh_nodes"] > info["n_bih_nodes"] else 16

    def algo_bytes(s):
        return (s["rays_primary"] + s["rays_secondary"]) * 64 + s["rays_shadow"] * 36 + (s["bih_nodes"] * 16 + s["mesh_nodes"] * 64) + s["prim_tests"] * s_prim

    kavg_ms = float(np.mean(kms[:nk])) if nk > 0 else float("nan")
    bytes_ref = algo_bytes(stf) * sf.G  # a launch carries sf.G frames of this rank's tiles
    achieved = bytes_ref / (kavg_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(HERE, "profiles", "pmc_traffic.json")  # written from a separate rocprofv3 --pmc run (profiles/README.md)
    if os.path.exists(pmc):
        try:
            per_frame = json.load(open(pmc)).get(args.scene, {}).get("hbm_bytes_per_frame")
            traffic = int(per_frame * sf.G) if per_frame else None  # per launch, like `achieved`
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "kernel": "k_render_flat" if info["tier"] == 0 else "k_render_generic", "kernel_ms_avg": round(kavg_ms, 4), "launches_timed": int(nk),
        "algorithmic_bytes_per_launch": int(bytes_ref),
        "per_ray": {"nodes": round((stf["bih_nodes"] + stf["mesh_nodes"]) / max(1, sum(rays) // world), 2), "prims": round(stf["prim_tests"] / max(1, sum(rays) // world), 2)},
        "visited_bytes_per_launch_early_out": int(algo_bytes(stc)) * sf.G, "frames_per_launch": sf.G, "timed": f"every {max(1, args.time_every)}th launch of the timed region",
        "frac_of_measured_copy_ceiling_6290GBs": round(achieved / 6290.0, 4),
        # with several frames in flight the launches overlap, so each launch's own duration (above) is longer than the
        # frame period; the same bytes over the measured frame period:
        "effective_GBs_over_frame_period": round(bytes_ref / sf.G / (ms_per_step * 1e-3) / 1e9, 1),
    }

    # ---- cpu_baseline: the oracle on a bounded sample of the same frame (every 4th tile), all host cores ----
    cpu = None
    if not args.no_cpu:
        sys.path.insert(0, os.path.join(HERE, "tests"))
        from oracle import oracle_py as O  # the checker, timed as the CPU baseline (kind "port")
        o, onmap, _ = O.load_scene(sd)
        o.set_camera_vectors(list(cam.pos), list(cam.fwd), list(cam.up), list(cam.right))
        # threads actually used: this process's CPU share, capped at the 16 host cores a one-GPU box is given
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 16))
        # the whole frame per repetition; repeat to >= 3 s of wall time (>= 3 repetitions) and report the median
        rates, cpu_rays, wall = [], 0, 0.0
        while len(rates) < 3 or wall < 3.0:
            tc = time.perf_counter()
            _, _, oc = o.render(W, H, mode=args.mode, maxdepth=maxdepth, nthreads=cores, want_packed=False)
            dt = time.perf_counter() - tc
            cpu_rays = oc["rays_primary"] + oc["rays_shadow"] + oc["rays_secondary"]
            rates.append(cpu_rays / dt / 1e6)
            wall += dt
            if len(rates) >= 40:
                break
        rate = float(np.median(rates))
        # one thread (the reference's `+RTS -N1`) on every 8th tile
        tc = time.perf_counter()
        _, _, o1 = o.render(W, H, mode=args.mode, maxdepth=maxdepth, tile_first=0, tile_stride=8, nthreads=1, want_packed=False)
        dt1 = time.perf_counter() - tc
        rate1 = (o1["rays_primary"] + o1["rays_shadow"] + o1["rays_secondary"]) / dt1 / 1e6
        cpu = {"value": round(rate, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": f"the same {W}x{H} frame, all 65x65 tiles ({cpu_rays} rays per repetition), {len(rates)} repetitions, median; {wall:.1f} s wall on {cores} threads; "
                         "fp64 C++ restatement of the reference algorithm (oracle/), tiles handed out dynamically like parMap",
               "seconds": round(wall, 2), "value_1_thread": round(rate1, 4), "gpu_over_cpu": round(value / rate, 1)}

    out = {
        "metric": "Mrays/sec + fps at 1920x1080 primary+shadow; 1/2/4/8 MI355X" if (W, H) == (1920, 1080) else f"Mrays/sec + fps at {W}x{H}",
        "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "fps": round(1e3 / ms_per_step, 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene}: " + {"S3": "100,352-triangle heightfield under bih (BASELINE configs[2]), 1 light, primary + shadow rays, maxdepth 1",
                                                     "S3mesh": "100,352-triangle heightfield as mesh (2-box BVH)", "S5": "1,002,528-triangle heightfield under bih"}.get(args.scene, args.scene),
                   "width": W, "height": H, "rays_per_frame": {"primary": rays[0], "shadow": rays[1], "secondary": rays[2]},
                   "sampling": "renderTile, 1 primary ray/pixel" if args.mode == 0 else "renderTileSubsample (adaptive, 1/8..2 primary rays/pixel)", "launches_in_flight": sf.n, "frames_per_launch": sf.G, "frame_product": "packed 0x00RRGGBB framebuffer (trace + blitTile fused, 4 B/pixel)" if args.product == "packed" else "float (r,g,b,a,depth) per pixel, 20 B/pixel", "tiles": f"{'64x64 work' if args.mode == 0 else '65x65 reference'} tiles, round-robin over ranks; a launch renders a rank's tiles of {sf.G} frames, one RCCL gather to rank 0 per launch, overlapped with the next launch" if (world > 1 or args.force_dist) else ("65x65 reference tile map, one GPU; a whole renderTile frame is cut into 64x64 work tiles (same pixels, no leftover strips)" if args.mode == 0 else "65x65 reference tiles, one GPU"),
                   "scene_setup_s": round(setup_s, 2), "bih_build": "host" if args.host_build else "device (glome_sb_bih_dev / glome_sb_mesh_dev) for lists of 4096+ objects", "device_bytes": info["device_bytes"]},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if dist_on:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
