"""The N > 1 path on CPU: world_size 2, gloo.  Each rank renders only the reference tiles it owns (round-robin,
Glome.hs:379-386 order), packs them into the dense tile payload, the ranks exchange with the plan's single gather, and
rank 0 blits -- the reassembled frame must equal the single-process frame bit for bit.  The renderer standing in for
the GPU here is the oracle; the shard plan, payload layout, gather and blit are the product's (glome_amd/dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 200, 150  # 4 x 3 tiles: ragged right / bottom tiles (5 and 20 pixels)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import oracle_for
    from glome_amd import api, dist, scenes
    sd = scenes.s1(nlights=1)
    o, om, _ = oracle_for(sd)
    P = api.render_params(width=W, height=H, maxdepth=1)
    plan = dist.ShardPlan(P, rank, world)
    mine, _, cnt = o.render(W, H, maxdepth=1, tile_first=rank, tile_stride=world, want_packed=False)  # untouched tiles stay 0
    lay = plan.layout(rank)
    payload = torch.zeros(plan.maxp, dtype=torch.float64)
    packed = dist.pack_numpy(mine, lay)
    assert packed.size == plan.sizes[rank]
    payload[:packed.size] = torch.from_numpy(packed)
    gathered = torch.zeros((world, plan.maxp), dtype=torch.float64) if rank == 0 else None
    plan.gather(payload, gathered)
    rays = torch.tensor([cnt["rays_primary"], cnt["rays_shadow"]], dtype=torch.float64)
    tdist.all_reduce(rays)
    if rank == 0:
        frame = np.full((H, W, 5), np.nan)
        for r in range(world):
            dist.blit_numpy(frame, gathered[r].numpy(), plan.layout(r))
        np.save(os.path.join(outdir, "frame.npy"), frame)
        np.save(os.path.join(outdir, "rays.npy"), rays.numpy())
    # the packed-pixel product (what bench.py gathers): one 0x00RRGGBB word per pixel, unit = 1
    plan1 = dist.ShardPlan(P, rank, world, unit=1)
    assert plan1.sizes[rank] * 5 == plan.sizes[rank] and plan1.maxp * 5 == plan.maxp
    _, mine_px, _ = o.render(W, H, maxdepth=1, tile_first=rank, tile_stride=world, want_packed=True)
    pk1 = dist.pack_numpy(mine_px.astype(np.int64), lay)
    pay1 = torch.zeros(plan1.maxp, dtype=torch.int64)
    pay1[:pk1.size] = torch.from_numpy(pk1)
    gat1 = torch.zeros((world, plan1.maxp), dtype=torch.int64) if rank == 0 else None
    plan1.gather(pay1, gat1)
    if rank == 0:
        fpx = np.full((H, W), -1, np.int64)
        for r in range(world):
            dist.blit_numpy(fpx, gat1[r].numpy(), plan1.layout(r))
        np.save(os.path.join(outdir, "frame_packed.npy"), fpx)
    # the pipelined path (FramePipeline): frames with different lights, in groups, several groups in flight
    frames = []

    def render_group(slot, payload_t, views):
        rows = payload_t.view(-1, plan.maxp)
        for g, k in enumerate(views):
            o.clear_lights()
            o.add_light([-100 + 40 * k, 70, 140], [7000, 5600, 5600])
            img, _, _ = o.render(W, H, maxdepth=1, tile_first=rank, tile_stride=world, want_packed=False)
            pk = dist.pack_numpy(img, lay)
            rows[g][:pk.size] = torch.from_numpy(pk)

    def blit(slot, g, gathered_t):
        f = np.full((H, W, 5), np.nan)
        for r in range(world):
            dist.blit_numpy(f, gathered_t[r].numpy()[g * plan.maxp:(g + 1) * plan.maxp], plan.layout(r))
        frames.append(f)

    # one frame per group, three groups in flight
    pipe = dist.FramePipeline(plan, [torch.zeros(plan.maxp, dtype=torch.float64) for _ in range(3)],
                              [torch.zeros((world, plan.maxp), dtype=torch.float64) if rank == 0 else None for _ in range(3)], render_group, blit)
    for k in range(5):
        pipe.step(k)
    pipe.flush()
    assert pipe.done == 5
    # two frames per launch and per collective, two groups in flight, 5 more frames (the last group is partial)
    G = 2
    pipe2 = dist.FramePipeline(plan, [torch.zeros(G * plan.maxp, dtype=torch.float64) for _ in range(2)],
                               [torch.zeros((world, G * plan.maxp), dtype=torch.float64) if rank == 0 else None for _ in range(2)], render_group, blit, group=G)
    for k in range(5, 10):
        pipe2.step(k)
    pipe2.flush()
    assert pipe2.done == 5 and pipe2.k == 3
    if rank == 0:
        np.save(os.path.join(outdir, "pipe_frames.npy"), np.stack(frames))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_tile_sharding_reassembles_the_frame(built, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    frame = np.load(tmp_path / "frame.npy")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_for
    from glome_amd import scenes
    o, om, _ = oracle_for(scenes.s1(nlights=1))
    whole, whole_px, cnt = o.render(W, H, maxdepth=1, want_packed=True)
    assert not np.isnan(frame).any()          # every pixel is owned by exactly one rank
    assert np.array_equal(frame, whole)       # bit exact
    assert np.array_equal(np.load(tmp_path / "frame_packed.npy"), whole_px.astype(np.int64))  # the packed framebuffer too
    assert np.load(tmp_path / "rays.npy").tolist() == [cnt["rays_primary"], cnt["rays_shadow"]]
    pf = np.load(tmp_path / "pipe_frames.npy")
    assert pf.shape[0] == 10
    for k in range(10):  # every pipelined frame (5 ungrouped, then 5 in groups of 2) equals its single-process render, in order
        o.clear_lights()
        o.add_light([-100 + 40 * k, 70, 140], [7000, 5600, 5600])
        ref, _, _ = o.render(W, H, maxdepth=1, want_packed=False)
        assert np.array_equal(pf[k], ref), k


def _weighted_worker(rank, world, port, outdir, pct):
    """three ranks, rank 0 with less than a fair share (glome_render_params.rank0_share_pct): payload sizes differ per rank,
    the gather pads to the largest"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import oracle_for
    from glome_amd import api, dist, scenes
    sd = scenes.s1(nlights=1)
    o, om, _ = oracle_for(sd)
    P = api.render_params(width=W, height=H, maxdepth=1, blocksize=16, rank0_share_pct=pct)  # 13 x 10 tiles
    plan = dist.ShardPlan(P, rank, world, unit=1)
    _, whole_px, _ = o.render(W, H, maxdepth=1, want_packed=True)  # (the oracle has no weighted shards: every rank packs its tiles of the whole frame)
    lay = plan.layout(rank)
    pk = dist.pack_numpy(whole_px.astype(np.int64), lay)
    assert pk.size == plan.sizes[rank]
    pay = torch.zeros(plan.maxp, dtype=torch.int64)
    pay[:pk.size] = torch.from_numpy(pk)
    gat = torch.zeros((world, plan.maxp), dtype=torch.int64) if rank == 0 else None
    plan.gather(pay, gat)
    if rank == 0:
        fpx = np.full((H, W), -1, np.int64)
        for r in range(world):
            dist.blit_numpy(fpx, gat[r].numpy(), plan.layout(r))
        np.save(os.path.join(outdir, "weighted.npy"), fpx)
        np.save(os.path.join(outdir, "weighted_sizes.npy"), np.array(plan.sizes))
    tdist.barrier()
    tdist.destroy_process_group()


def test_three_ranks_with_a_weighted_share_reassemble_the_frame(built, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_weighted_worker, args=(3, port, str(tmp_path), 60), nprocs=3, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_for
    from glome_amd import scenes
    o, om, _ = oracle_for(scenes.s1(nlights=1))
    _, whole_px, _ = o.render(W, H, maxdepth=1, want_packed=True)
    assert np.array_equal(np.load(tmp_path / "weighted.npy"), whole_px.astype(np.int64))
    sizes = np.load(tmp_path / "weighted_sizes.npy")
    assert sizes.sum() == W * H and sizes[0] < 0.8 * sizes[1] and abs(int(sizes[1]) - int(sizes[2])) <= 2 * 16 * 16


def test_shard_plan_partitions_tiles(built):
    from glome_amd import api, dist
    for (w, h, world) in [(720, 480, 8), (1920, 1080, 8), (3840, 2160, 8), (200, 150, 3), (64, 64, 2)]:
        P = api.render_params(width=w, height=h)
        seen = np.zeros((h, w), np.int32)
        total = 0
        for r in range(world):
            plan = dist.ShardPlan(P, r, world)
            lay = plan.layout(r)
            base = 0
            for x, y, tw, th, pb in lay:
                assert pb == base
                base += tw * th
                seen[y:y + th, x:x + tw] += 1
            assert base * 5 == plan.sizes[r]
            total += base
        assert total == w * h and (seen == 1).all()
    # balance: at 1080p over 8 ranks no rank carries more than ~4% above the mean
    P = api.render_params(width=1920, height=1080)
    sizes = dist.ShardPlan(P, 0, 8).sizes
    assert max(sizes) / (sum(sizes) / 8) < 1.06


def test_bench_cuts_a_run_into_equal_launches():
    """bench.py's frames_per_launch: as few launches as 32 frames a launch allow, of equal size, in whole rounds over the launches in
    flight once there are that many (a 200-step run: eight launches of 25, not six of 32 and one of 8; the driver's 20-step run: one)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    f = m.frames_per_launch
    assert [f(s, 4) for s in (1, 3, 4, 20, 32, 33, 64, 100, 128, 200, 1000)] == [1, 1, 4, 20, 32, 17, 32, 25, 32, 25, 32]
    for steps in range(4, 400):
        for lanes in (1, 2, 3, 4, 8):
            g = f(steps, lanes)
            n = -(-steps // g)
            assert 1 <= g <= 32 and n * g >= steps and (n - 1) * g < steps   # the run is covered, by launches no larger than a launch may be
