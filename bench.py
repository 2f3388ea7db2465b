#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mrays/s (and fps) of whole-frame rendering at 1920x1080,
primary + shadow rays, on the 100k-triangle BIH scene (BASELINE.json configs[2] = SURVEY.md scene S3).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one frame: every pixel's primary ray + its shadow rays through the hot path (raygen -> BIH closest hit ->
shadow any-hit -> shade), scene resident in HBM.  With N GPUs the frame's 65x65 reference tiles are sharded round-robin
and gathered to rank 0 with one RCCL gather (strong scaling: the frame is fixed).  Rank 0 prints ONE JSON line.

The line carries `roofline` and `cpu_baseline`.

roofline: the dominant kernel is a cache-resident tree walk, not a stream, so one number against HBM peak says little.  The
line therefore reports one fraction PER CEILING -- HBM / fabric bytes, L2 requests, scalar-unit issue, vector issue -- from
the counters of a committed `rocprofv3 --pmc` pass over the SAME launch shape (profiles/<round>_pmc_<scene>.json, made by
tools/pmc_roofline.py) and the duration of that launch measured live here with HIP events on the launch stream, launches
NOT overlapped (one in flight, 8 frames each: enough work to fill the GPU).  `bound` names the highest ceiling; `frac` is
its fraction (<= 1).  SURVEY.md 8(d)'s algorithmic byte model stays as a labelled side figure (`model`): it charges every
ray for every node a CPU would fetch, while a wave fetches a node once for its 64 rays.

cpu_baseline: the CPU oracle -- a C++ restatement of the reference algorithm, kind "port" -- timed on this box's host
cores on the same frame.
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# issue peaks of the scalar port per CU, measured (tools/probe/valu_rate.hip; profiles/r03_valu_rate.txt, r04_valu_rate_branches.txt)
SCALAR_PEAK = 0.95                # s_add_u32 alone: 0.236-0.237 per cycle per SIMD
# A branch does NOT cost the scalar port a full slot (round-4 probe, kinds 17-23 at six waves per SIMD): not-taken s_cbranch alone 0.247
# per cycle per SIMD, taken s_branch alone 0.149, but branch : s_add mixes issue 0.27-0.33 of BOTH kinds together -- more than s_add
# alone (0.237).  Scalar-type instructions with the kernel's share of branches (two in five) are priced against the mixes' 1.25 per CU.
SCALAR_PEAK_WITH_BRANCHES = 1.25


def frames_per_launch(steps, lanes, most=32):
    """A run of `steps` frames cut into as few launches as a launch's `most` frames allow, of equal size, and -- once there are as many
    launches as are kept in flight -- into a whole number of rounds over those."""
    if steps < 4:
        return 1
    n = -(-steps // most)
    if n >= lanes:
        n = -(-n // lanes) * lanes
    return -(-steps // n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="S3", help="S1 S2 S3 S3mesh S4 S5 S5mesh, TS = GlomeView's own default scene, TSnooak = the same without the oak (default: the headline workload S3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--mode", type=int, default=0, help="0 = renderTile (1 ray/pixel), 1 = renderTileSubsample (adaptive)")
    ap.add_argument("--lanes", type=int, default=None, help="launches kept in flight per GPU (HIP streams / context slots); default 4 (3 for 16-frame launches at 8+ ranks)")
    ap.add_argument("--orbit", type=float, default=0.25, help="degrees the camera moves around its look-at point from one frame to the next (a triangle wave over 32 frames, "
                                                               "so the frames in flight are different views); 0 = every frame the same view.  `value` is the orbit's; the line also "
                                                               "carries the fixed-camera period (`fixed_camera`)")
    ap.add_argument("--time-every", type=int, default=1, help="HIP-event pair on every k-th launch of the timed region (roofline kernel time)")
    ap.add_argument("--host-build", action="store_true", help="build the BIH with the host builder (glome_sb_bih) instead of on the GPU")
    ap.add_argument("--force-dist", action="store_true", help="one GPU, but through the multi-GPU pipeline with a one-rank RCCL group (rehearsal)")
    ap.add_argument("--group", type=int, default=0, help="frames per launch (and per RCCL gather); default by transport and run length (up to 32)")
    ap.add_argument("--product", default="packed", choices=["packed", "rgbad"],
                    help="what a frame is: GlomeView's framebuffer of packed 0x00RRGGBB pixels (blitTile; 4 B/pixel cross xGMI) "
                         "or the float (r,g,b,a,depth) tuples (20 B/pixel)")
    ap.add_argument("--transport", default="auto", choices=["auto", "direct", "gather"], help="several ranks: how the pixels reach rank 0 (auto: direct stores when the frames can be mapped, else the gather)")
    ap.add_argument("--rehearse", action="store_true", help="N ranks on ONE GPU over gloo (payloads staged through the host): a rehearsal of the multi-process "
                                                            "control flow where there is a single device; its timings mean nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks ourselves, as fresh child processes, BEFORE this process makes any GPU
        # call (it never does: it only waits and passes the children's output and exit code on)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GLOME_BENCH_SELF_LAUNCHED="1")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU path to time)")
    if args.rehearse:
        local = 0  # every rank on the one device
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.force_dist and world == 1:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse:
            tdist.init_process_group("gloo")
        else:
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

    regroup_for_gather = False

    def gather_group(steps):
        # several ranks, payloads gathered on rank 0 (tools/short_run_groups.py, profiles/r03_short_run_groups.log: rank 0's side of a short
        # run by frames per launch, one-GPU rehearsal): at 8 ranks a 20-step run takes 0.078 / 0.043 / 0.039 ms per step with 1 / 4 / 8 frames per
        # launch, a 100-step run 0.039 / 0.031 / 0.027 with 4 / 8 / 16 (tools/shard_timing.py, profiles/r03_h_shard_timing.log)
        return (1 if world == 2 else 4) if steps < 8 else (4 if steps < 16 else (8 if steps < 40 else 16))

    if args.group <= 0:
        # frames per launch (and per gather).  Deep batches pay a fill / drain of about one launch per run, so short runs get
        # shallower ones; the break-even points are measured (below)
        if args.product != "packed":
            args.group = 1
        elif args.mode != 0:
            # the adaptive sampler: a batch of eight or more frames lets it work in large regions (fewer, fuller sample packets),
            # and a tile's five dependent passes are one wave's work at a time -- the more tiles a launch holds the more waves are busy
            # (S3: 0.276 / 0.251 / 0.242 ms per frame with 8 / 12 / 16 frames per launch; profiles/r03_probes/adaptive_frames_per_launch.txt)
            args.group = (16 if args.steps >= 32 else (8 if args.steps >= 16 else 1)) if world == 1 else 1
        elif world == 1:
            # A launch cannot be shorter than its slowest work items (about 0.33 ms on S3), whatever it carries: the more frames share
            # it the better, up to the 16 a launch may carry, and a short run is best cut into two launches that are in flight
            # together (tools/short_run_sweep.sh, profiles/r03_short_run_sweep.log: 20 steps take 0.202 ms per step as 5 x 4 frames
            # and 0.188 as 2 x 10; 200 steps 0.1725 at 8 frames per launch and 0.162 at 16)
            # round 4: a launch costs ~0.31 ms besides its frames even with others in flight (profiles/r04_probes/short_run_fit.txt:
            # per launch 0.307 + 0.137 ms x frames, pipelined), so a run is cut into as few launches as possible: up to 32 frames each
            # ... and into launches of equal size, a whole number of rounds over the launches in flight (200 steps: eight launches of 25
            # rather than six of 32 and one of 8: 0.1433 against 0.146 ms per step, profiles/r04_probes/bench_lanes_groups_sweep.txt)
            args.group = frames_per_launch(args.steps, args.lanes or 4)
        elif args.transport != "gather":
            # several ranks, the direct transport (every rank's kernel stores into rank 0's frames): a rank's step is its launch and
            # nothing else, so the one-GPU rule holds -- a 20-step run at 8 ranks takes 0.63 ms as one launch of twenty shard frames, 0.73 as
            # two of ten, 0.98 as four of five (tools/probe/short_run_shards.py, profiles/r04_probes/short_run_shards.txt); should the frames
            # turn out not to be mappable, the gather's own rule (below) is applied to a pipeline built again
            args.group = frames_per_launch(args.steps, args.lanes or 4)
            regroup_for_gather = True
        else:
            args.group = gather_group(args.steps)
            if world >= 8 and args.group == 16 and args.lanes is None:
                args.lanes = 3
    lanes = args.lanes if args.lanes is not None else 4
    sf = dist.ShardedFrame(scene, P, rank, world, device, lanes=lanes, product=args.product, group=args.group, force_pipeline=args.force_dist,
                           direct={"auto": None, "direct": True, "gather": False}[args.transport])
    if regroup_for_gather and not sf.direct:  # (every rank takes this branch or none: the mode was agreed by an all-reduce)
        sf.close()
        args.group = gather_group(args.steps)
        if world >= 8 and args.group == 16 and args.lanes is None:
            lanes = 3
        sf = dist.ShardedFrame(scene, P, rank, world, device, lanes=lanes, product=args.product, group=args.group, force_pipeline=args.force_dist, direct=False)
    args.lanes = lanes

    def barrier():
        if dist_on:
            tdist.barrier()
        torch.cuda.synchronize(device)

    # The views: the scene's own camera, or (--orbit, the default) that camera moved around its look-at point by a fixed angle per
    # frame -- a triangle wave over 32 frames -- so that the frames a launch carries, and the launches in flight, are different views
    # (identical frames share every scalar-cache and L2 line: VERDICT r03).  A frame's rays are counted per view, before the timed region.
    def orbit_view(k):
        if not args.orbit:
            return cam
        pos, at, up, ang = sd.cam
        th = np.deg2rad(args.orbit) * (k % 32 if k % 32 <= 16 else 32 - k % 32)
        p, a = np.asarray(pos, np.float64), np.asarray(at, np.float64)
        d = p - a
        rot = np.array([d[0] * np.cos(th) + d[2] * np.sin(th), d[1], -d[0] * np.sin(th) + d[2] * np.cos(th)])
        return api.camera(tuple(a + rot), at, up, ang)
    n_views = 17 if args.orbit else 1
    views = [orbit_view(k) for k in range(32)] if args.orbit else [cam]

    def count_rays(view):
        st = sf.step(view, lights, stats=True)
        if dist_on:
            t_ = torch.tensor([st["rays_primary"], st["rays_shadow"], st["rays_secondary"]], dtype=torch.float64, device=device)
            tdist.all_reduce(t_)
            return [int(x) for x in t_.tolist()]
        return [st["rays_primary"], st["rays_shadow"], st["rays_secondary"]]
    rays_of_view = [count_rays(views[k]) for k in range(n_views)]
    rays_by_k = [rays_of_view[k % 32 if k % 32 <= 16 else 32 - k % 32] if args.orbit else rays_of_view[0] for k in range(32)]
    rays = rays_of_view[0]
    rays_fixed = sum(rays)

    sf.prime(cam, lights)  # every lane's slot / stream / kernel instance exists before the warm-up (initialisation, not steps)

    def timed_region(view_of_step):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks"""
        for k in range(args.warmup):
            sf.step(view_of_step(k), lights)
        sf.flush()
        barrier()
        # every --time-every-th launch of the timed region carries a HIP-event pair on its launch stream (default: every launch)
        ctx.lib.glome_ctx_timing_begin_sampled(ctx.h, args.steps, max(1, args.time_every))
        t_start = time.perf_counter()
        for k in range(args.steps):
            sf.step(view_of_step(args.warmup + k), lights)
        sf.flush()  # frames are pipelined (several in flight): complete the last ones inside the timed region
        barrier()
        dt = time.perf_counter() - t_start
        kms = np.zeros(args.steps, np.float32)
        ctx.lib.glome_ctx_timing_end(ctx.h, kms.ctypes.data_as(L.c_fp), args.steps)
        if dist_on:
            e = torch.tensor([dt], dtype=torch.float64, device=device)
            tdist.all_reduce(e, op=tdist.ReduceOp.MAX)
            dt = float(e.item())
        return dt
    elapsed_fixed = timed_region(lambda k: cam) if args.orbit else None  # the same view every frame (what rounds 1-3 timed)
    elapsed = timed_region(lambda k: views[k % 32] if args.orbit else cam)
    last_view = views[(args.warmup + args.steps - 1) % 32] if args.orbit else cam
    rays_timed = sum(sum(rays_by_k[(args.warmup + k) % 32]) for k in range(args.steps)) if args.orbit else rays_fixed * args.steps

    if rank != 0:
        if dist_on:
            sf.close()
            tdist.barrier()
            tdist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = rays_timed / elapsed / 1e6  # the rays of the frames actually rendered in the timed region
    rays_per_step = rays_timed / args.steps

    # ---- after the timed region: the last frame the job delivered (rank 0's framebuffer, reassembled from every rank's tiles)
    # against the same view rendered whole on this GPU -- the multi-GPU data path checks itself in every run
    frame_check = None
    if args.product == "packed":
        torch.cuda.synchronize(device)
        ctx.lib.glome_ctx_use_slot(ctx.h, None, 0)
        whole = torch.zeros((H, W), dtype=torch.int32, device=device)
        scene.render_dev(last_view, lights, P, None, whole.data_ptr(), want_stats=False)
        ctx.synchronize()
        frame_check = bool(torch.equal(sf.frame.reshape(H, W), whole))

    # ---- the dominant kernel alone: launch duration and single-frame latency (HIP events on the launch stream) ----
    # One launch in flight at a time (the timed region above overlaps several, so a launch's own duration there is longer
    # than its share of the GPU).  G_lone frames per launch: enough work items to fill every CU many times over.
    la = (L.Light * max(1, len(lights)))(*lights)
    ctx.lib.glome_ctx_use_slot(ctx.h, None, 0)
    torch.cuda.synchronize(device)

    def lone_launches(nframes, reps):
        """ms per launch of `reps` launches carrying `nframes` frames of this rank's tiles each, never overlapped"""
        cams = (L.Camera * nframes)(*([cam] * nframes))
        if world == 1:
            buf = torch.zeros((nframes, H, W), dtype=torch.int32, device=device)
            call = lambda: ctx.lib.glome_render_packed_batch_dev(scene.h, cams, nframes, la, len(lights), C.byref(sf.P), C.c_void_p(buf.data_ptr()), H * W, None)
        else:
            buf = torch.zeros(nframes * sf.plan.maxp, dtype=torch.int32, device=device)
            call = lambda: ctx.lib.glome_render_tiles_packed_batch_dev(scene.h, cams, nframes, la, len(lights), C.byref(sf.P_local), C.c_void_p(buf.data_ptr()), sf.plan.maxp, None)
        for _ in range(2):
            if call() != 0:
                raise SystemExit("lone launch: " + ctx.err())
            ctx.synchronize()
        ctx.lib.glome_ctx_timing_begin(ctx.h, reps)
        for _ in range(reps):
            if call() != 0:
                raise SystemExit("lone launch: " + ctx.err())
            ctx.synchronize()
        out = np.zeros(reps, np.float32)
        n = ctx.lib.glome_ctx_timing_end(ctx.h, out.ctypes.data_as(L.c_fp), reps)
        return [float(x) for x in out[:n]]

    G_lone = 8
    ctx.lib.glome_ctx_set_grid_per_cu(ctx.h, 32)  # a launch on its own takes every wave slot its registers / LDS allow
    lone_ms = lone_launches(G_lone, 12)
    ctx.lib.glome_ctx_set_grid_per_cu(ctx.h, 0)
    # one frame alone: the library's own sizing of a launch that has the GPU to itself (a flagship frame takes 0.40 ms with the 12 waves per CU
    # it picks, 0.51 with every slot filled: profiles/r04_probes/lone_launch_grid.txt) -- what a caller of glome_render gets
    one_ms = lone_launches(1, 12) if G_lone != 1 else lone_ms
    kernel_ms = float(np.median(lone_ms))
    kernel_s = kernel_ms * 1e-3
    latency = {"single_frame_ms": round(float(np.median(one_ms)), 4), "lone_launch_ms": round(kernel_ms, 4), "lone_launch_frames": G_lone,
               "ms_per_frame_in_a_lone_launch": round(kernel_ms / G_lone, 4), "pipelined_ms_per_frame": round(ms_per_step, 4),
               "fixed_camera_ms_per_step": round(elapsed_fixed / args.steps * 1e3, 4) if elapsed_fixed else None,
               "note": "value / ms_per_step are pipelined throughput (launches_in_flight x frames_per_launch independent frames in flight); single_frame_ms is one frame alone on an idle GPU"}

    # ---- SURVEY.md 8(d)'s byte model, kept as a labelled side figure ----
    # 32 B ray in + 32 B hit out per closest-hit ray, 32 + 4 per shadow ray, 16 B per BIH node visited (64 B per Mesh
    # node), S_prim per primitive tested (48 B triangle, 16 B sphere); visits as the reference algorithm makes them per
    # ray (no early-out, rayint_debug convention, Bih.hs:378-412), counted by the faithful kernel on this very frame and
    # these very tiles (sf.P_local; tests pin the counts to the CPU oracle's).
    tmp = torch.zeros((H, W, 5), dtype=torch.float32, device=device)
    torch.cuda.synchronize(device)
    stf = scene.render_dev(cam, lights, dist._clone_params(sf.P_local, faithful=1, count_work=1), tmp.data_ptr())
    stc = scene.render_dev(cam, lights, dist._clone_params(sf.P_local, faithful=0, count_work=1), tmp.data_ptr())
    del tmp
    s_prim = 48 if info["n_triangles"] >= info["n_spheres"] else 16

    def algo_bytes(s_):
        return (s_["rays_primary"] + s_["rays_secondary"]) * 64 + s_["rays_shadow"] * 36 + (s_["bih_nodes"] * 16 + s_["mesh_nodes"] * 64) + s_["prim_tests"] * s_prim

    rays_local_n = max(1, stf["rays_primary"] + stf["rays_shadow"] + stf["rays_secondary"])
    model = {"what": "SURVEY.md 8(d) algorithmic bytes: every ray charged for every node / primitive the reference's per-ray traversal fetches; NOT bytes this kernel moves (a wave fetches a node once for its 64 rays, from L2)",
             "bytes_per_launch": int(algo_bytes(stf)) * G_lone, "bytes_per_frame": int(algo_bytes(stf)),
             "per_ray": {"nodes": round((stf["bih_nodes"] + stf["mesh_nodes"]) / rays_local_n, 2), "prims": round(stf["prim_tests"] / rays_local_n, 2)},
             "bytes_per_frame_early_out_visits": int(algo_bytes(stc)),
             "GBs_over_lone_launch": round(algo_bytes(stf) * G_lone / kernel_s / 1e9, 1)}

    # ---- per-ceiling fractions from the committed counter pass of this launch shape ----
    _, cus, _ = ctx.device_info()
    kname = ("k_ss_frame_flat" if args.mode != 0 else "k_render_flat") if info["tier"] == 0 else ("k_ss_frame_generic" if args.mode != 0 else "k_render_generic")
    roofline = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "kernel": kname, "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_all": [round(x, 4) for x in lone_ms], "launch_shape": f"1 launch in flight, {G_lone} frame(s) per launch, rank 0's tiles", "ceilings": None,
                "pmc_source": None, "model": model}
    # the newest committed counter pass of this scene and mode, by round number (r9 < r10), made from THESE kernel sources
    import re
    cands = sorted(glob.glob(os.path.join(HERE, "profiles", f"r*_pmc_{args.scene}_mode{args.mode}.json")),
                   key=lambda p: (int(re.match(r"r(\d+)", os.path.basename(p)).group(1)), os.path.basename(p)))
    if cands and world == 1:
        try:
            sys.path.insert(0, os.path.join(HERE, "tools"))
            from pmc_roofline import source_sha16
            pm = json.load(open(cands[-1]))
            roofline["pmc_source"] = os.path.relpath(cands[-1], HERE)
            if pm.get("source_sha16") and pm["source_sha16"] != source_sha16(HERE):
                # counters of another build say nothing about this one: no fractions rather than stale ones
                roofline.update({"stale": True, "note": "the committed counter pass was made from other kernel sources than the library timed here: re-run tools/pmc_roofline.sh"})
                raise StopIteration
            cn = pm["counters"]
            sc_ = G_lone / float(pm["frames_per_launch"])  # counts are proportional to the frames a launch carries
            clock_hz = float(pm.get("clock_ghz") or 2.35) * 1e9
            cyc = kernel_s * clock_hz  # cycles of one CU over the launch
            hbm_bytes = (cn["FETCH_SIZE"] * 2.0 + cn["WRITE_SIZE"]) * 1024.0 * sc_  # KiB; gfx950 FETCH_SIZE reads half (MI355X_MICROARCH.md, HBM)
            l2_bytes = cn["TCC_REQ_sum"] * 128.0 * sc_                               # an upper bound: every request priced as a full 128-B line
            sal_nb = (cn["SQ_INSTS_SALU"] + cn["SQ_INSTS_SMEM"]) * sc_
            sal = sal_nb + cn["SQ_INSTS_BRANCH"] * sc_
            allin = sal + (cn["SQ_INSTS_VALU"] + cn.get("SQ_INSTS_LDS", 0.0) + cn.get("SQ_INSTS_VMEM_RD", 0.0) + cn.get("SQ_INSTS_VMEM_WR", 0.0)) * sc_
            # issue peaks measured on this GPU (tools/probe/valu_rate.hip, profiles/r03_valu_rate.txt): a SIMD issues 0.236 scalar
            # instructions per cycle, 0.32-0.40 vector instructions on vector operands (0.24 with a scalar operand, 0.19-0.23 packed),
            # and 0.56 instructions of all kinds in the probe's best mix (two vector : one scalar; 0.43 alternating); x 4 SIMDs per CU.
            # The scalar and the vector port overlap, so "all kinds" is the loosest of the three issue ceilings, not their sum.
            ceil = {
                "hbm": {"achieved": round(hbm_bytes / kernel_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_bytes / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "fabric-side bytes (FETCH_SIZE x 2 + WRITE_SIZE); Infinity-Cache hits included, so true HBM traffic is at most this"},
                "l2": {"achieved": round(l2_bytes / kernel_s / 1e9, 1), "peak": 34500.0, "unit": "GB/s", "frac": round(l2_bytes / kernel_s / 1e9 / 34500.0, 4)},
                # scalar-type issue, three ways so rounds stay comparable.  What a branch costs the scalar port is
                # measured, not assumed (tools/probe/valu_rate.hip kinds 17-23, profiles/r04_valu_rate_branches.txt): SCALAR_PEAK_WITH_BRANCHES
                # is the rate of the probe's branch : s_add mixes, SCALAR_PEAK that of s_add alone.
                "scalar_issue": {"achieved": round(sal / (cus * cyc), 4), "peak": SCALAR_PEAK_WITH_BRANCHES, "unit": "scalar-type (SALU + SMEM + branch) instructions per cycle per CU",
                                 "frac": round(sal / (cus * cyc) / SCALAR_PEAK_WITH_BRANCHES, 4)},
                "scalar_issue_without_branches": {"achieved": round(sal_nb / (cus * cyc), 4), "peak": SCALAR_PEAK, "unit": "SALU + SMEM instructions per cycle per CU, against the measured s_add rate",
                                                  "frac": round(sal_nb / (cus * cyc) / SCALAR_PEAK, 4)},
                "scalar_issue_r02_definition": {"achieved": round(sal_nb / (cus * cyc), 4), "peak": 1.0, "unit": "SALU + SMEM per cycle per CU over a nominal 1.0 (round 2's definition, kept so rounds compare)",
                                                "frac": round(sal_nb / (cus * cyc), 4)},
                "valu_issue": {"achieved": round(cn["SQ_INSTS_VALU"] * sc_ / (cus * cyc), 4), "peak": 1.6, "unit": "wave64 vector instructions per cycle per CU (the best rate the probe measured: VOP2 on vector operands; one with a scalar operand issues at 0.96, a packed one at 0.75-0.9)",
                               "frac": round(cn["SQ_INSTS_VALU"] * sc_ / (cus * cyc) / 1.6, 4),
                               "frac_of_guide_peak": round(cn["SQ_INSTS_VALU"] * sc_ / (cus * cyc) / 2.0, 4), "guide_peak": 2.0,
                               "guide_peak_note": "MI355X_MICROARCH.md: 4 SIMDs x one wave64 VALU instruction per 2 cycles; this kernel's vector instructions carry scalar operands (node planes, triangle words), which the probe issues at 0.96"},
                "issue_all": {"achieved": round(allin / (cus * cyc), 4), "peak": 2.25, "unit": "instructions of all kinds per cycle per CU (the best mix the probe measured: two vector on vector operands to one scalar)", "frac": round(allin / (cus * cyc) / 2.25, 4)},
            }
            top = max((k for k in ceil if k not in ("scalar_issue_without_branches", "scalar_issue_r02_definition")), key=lambda k: ceil[k]["frac"])
            wc = cn.get("SQ_WAVE_CYCLES")
            roofline.update({"bound": top, "achieved": ceil[top]["achieved"], "peak": ceil[top]["peak"], "unit": ceil[top]["unit"], "frac": ceil[top]["frac"],
                             "traffic": int(hbm_bytes), "ceilings": ceil, "clock_ghz": pm.get("clock_ghz"),
                             "hbm_frac": ceil["hbm"]["frac"],
                             "hbm_frac_note": "fabric-side bytes of the launch (FETCH_SIZE x 2 + WRITE_SIZE, counters) / its duration / 8 TB/s: the north star's 'fraction of the HBM roofline'; low by design -- rays live in registers and the scene in cache",
                             "model_frac": round(model["bytes_per_launch"] / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                             "model_frac_note": "SURVEY.md 8(d)'s algorithmic bytes (every ray charged for every node and triangle the reference's per-ray traversal fetches) / duration / 8 TB/s; ABOVE 1 because a wave fetches a node once for its 64 rays: those bytes are work done, not bytes moved",
                             "wave_wait_frac": round(cn["SQ_WAIT_ANY"] / wc, 4) if wc else None,
                             "wave_issue_stall_frac": round(cn["SQ_WAIT_INST_ANY"] / wc, 4) if wc and cn.get("SQ_WAIT_INST_ANY") else None,
                             "scalar_cache_hit": round(cn["SQC_DCACHE_HITS"] / (cn["SQC_DCACHE_HITS"] + cn["SQC_DCACHE_MISSES"]), 4) if cn.get("SQC_DCACHE_HITS") else None,
                             "lane_utilisation": round(cn["SQ_THREAD_CYCLES_VALU"] / (cn["SQ_ACTIVE_INST_VALU"] * 64.0), 4) if cn.get("SQ_THREAD_CYCLES_VALU") and cn.get("SQ_ACTIVE_INST_VALU") else None,
                             "pmc_kernel_ms": pm.get("kernel_ms"),
                             "reading": "no issue port is saturated (the busiest: %s at %.2f) and no memory level is near its bandwidth: the waves wait (wave_wait_frac) and execute their own instruction streams serially -- the frame time follows the instruction count one to one (DESIGN.md 4.1a, 4.6)" % (top, ceil[top]["frac"]),
                             "note": "counters per launch from the committed rocprofv3 --pmc passes of this launch shape; duration measured live (HIP events); cycles = duration x the clock the counter passes measured"})
        except StopIteration:
            pass
        except Exception as e:  # a malformed profile must not void the bench line
            roofline["pmc_error"] = repr(e)

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
                                                     "S3mesh": "100,352-triangle heightfield as mesh (2-box BVH)", "S5": "1,002,528-triangle heightfield under bih", "S5mesh": "1,002,528-triangle heightfield as mesh (2-box BVH; a Mesh casts no shadows, Q12)",
                                                     "TS": "GlomeView's default scene geom'' (TestScene.hs:183-197) without the oak, at GlomeView's 720x480, maxdepth 3"}.get(args.scene, args.scene),
                   "width": W, "height": H, "rays_per_frame": {"primary": rays[0], "shadow": rays[1], "secondary": rays[2]},
                   "sampling": "renderTile, 1 primary ray/pixel" if args.mode == 0 else "renderTileSubsample (adaptive, 1/8..2 primary rays/pixel)", "launches_in_flight": sf.n, "frames_per_launch": sf.G, "rank0_share_pct": sf.rank0_share_pct,
                   "transport": ("direct: every rank's kernel stores its tiles' pixels straight into rank 0's frames (HIP IPC mapping); one one-word all-reduce per launch as the completion signal" if sf.direct else ("gather: tile payloads to rank 0 with one RCCL gather per launch, blitted there" + (" (direct stores were not possible: %s)" % sf.direct_error if getattr(sf, "direct_error", None) else ""))) if (world > 1 or args.force_dist) else "none (one GPU)", "frame_product": "packed 0x00RRGGBB framebuffer (trace + blitTile fused, 4 B/pixel)" if args.product == "packed" else "float (r,g,b,a,depth) per pixel, 20 B/pixel", "tiles": f"{'64x64 work' if args.mode == 0 else '65x65 reference'} tiles, round-robin over ranks; a launch renders a rank's tiles of {sf.G} frames, one RCCL gather to rank 0 per launch, overlapped with the next launch" if (world > 1 or args.force_dist) else ("65x65 reference tile map, one GPU; a whole renderTile frame is cut into 64x64 work tiles (same pixels, no leftover strips)" if args.mode == 0 else "65x65 reference tiles, one GPU"),
                   "scene_setup_s": round(setup_s, 2), "bih_build": "host" if args.host_build else "device (glome_sb_bih_dev / glome_sb_mesh_dev) for lists of 4096+ objects", "device_bytes": info["device_bytes"]},
        "camera": ({"orbit_deg_per_frame": args.orbit, "views": 32, "note": "`value` / `ms_per_step`: the camera moves around its look-at point by this angle per frame (triangle wave over 32 frames), rays counted per view; `fixed_camera`: every frame the same view, as rounds 1-3 timed"} if args.orbit else {"orbit_deg_per_frame": 0}),
        "fixed_camera": ({"ms_per_step": round(elapsed_fixed / args.steps * 1e3, 4), "value": round(rays_fixed * args.steps / elapsed_fixed / 1e6, 2), "unit": "Mrays/s"} if elapsed_fixed else None),
        "roofline": roofline, "latency": latency, "cpu_baseline": cpu, "frame_equals_single_gpu_render": frame_check, "rccl_ranks": world if (dist_on and not args.rehearse) else 0,
        **({"rehearsal": "ranks share one GPU, payloads gathered over gloo through the host: not a measurement"} if args.rehearse else {}),
    }
    print(json.dumps(out), flush=True)
    if dist_on:
        sf.close()
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
