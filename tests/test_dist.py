"""Multi-GPU sharding logic without GPUs: tile ownership and the one reduce (rayzen_amd/dist.py), rehearsed
with world_size-2 gloo processes that use the CPU oracle as the per-rank renderer."""
import os
import socket
import sys

import numpy as np
import pytest

from rayzen_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("W,H,n", [(1920, 1080, 8), (256, 256, 2), (100, 37, 3), (8, 8, 4), (1, 1, 2)])
def test_owner_map_partitions_the_frame(W, H, n):
    own = D.owner_map(W, H, n)
    assert own.shape == (H, W) and own.min() >= 0 and own.max() < n
    tx, ty = D.tile_grid(W, H)
    for y in range(0, H, 8):                      # constant inside a tile, round-robin across tiles
        for x in range(0, W, 8):
            t = (y // 8) * tx + x // 8
            assert (own[y:y + 8, x:x + 8] == t % n).all()
    assert sum(D.local_tile_count(W, H, r, n) for r in range(n)) == tx * ty
    assert sum(D.owned_samples(W, H, 3, r, n) for r in range(n)) == W * H * 3


def test_1080p_split_is_balanced_over_8_ranks():
    own = D.owner_map(1920, 1080, 8)
    counts = np.bincount(own.ravel(), minlength=8)
    assert counts.max() - counts.min() <= 64 * 2           # within two tiles of each other


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, H, spp, bounces, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import rzo
    from rayzen_amd import scene as S
    from helpers import oracle_frame, oracle_scene
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = S.cornell_scene()
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, spp, bounces)
    own = D.owner_map(W, H, world)
    acc = np.zeros((H, W, 4), np.float32)                   # zero outside this rank's tiles
    tx, ty = D.tile_grid(W, H)
    for t in range(rank, tx * ty, world):                   # the same tile deal the kernel uses
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        rzo.render(osc, fr, accum=acc, crop=(x0, y0, min(x0 + 8, W), min(y0 + 8, H)), nthreads=1)
    assert (acc[own != rank] == 0).all()
    ten = torch.from_numpy(acc)
    D.reduce_accum(ten, dst=0)
    if rank == 0:
        np.save(out_path, ten.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharded_render_plus_reduce_is_bit_identical_to_single_rank(tmp_path):
    import torch.multiprocessing as mp
    from helpers import oracle_render
    from rayzen_amd import scene as S
    W, H, spp, bounces = 72, 40, 2, 3
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, spp, bounces, out), nprocs=2, join=True)
    got = np.load(out)
    want = oracle_render(S.cornell_scene(), W, H, spp, bounces, nthreads=2)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def _bench_like_worker(rank, world, port, out_path):
    """The step() of bench.py with the oracle standing in for the GPU: private buffer -> copy -> in-place reduce,
    repeated, must give the same frame every step (an in-place reduce straight on the render buffer would not)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import rzo
    from rayzen_amd import scene as S
    from helpers import oracle_frame, oracle_scene
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 40, 24
    sc = S.cornell_scene()
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, 2, 2)
    tx, ty = D.tile_grid(W, H)
    accum = torch.zeros((H, W, 4), dtype=torch.float32)
    frame = torch.empty_like(accum)
    frames = []
    for step in range(3):
        a = accum.numpy()
        for t in range(rank, tx * ty, world):
            x0, y0 = (t % tx) * 8, (t // tx) * 8
            rzo.render(osc, fr, accum=a, crop=(x0, y0, min(x0 + 8, W), min(y0 + 8, H)), nthreads=1)
        frame.copy_(accum)
        D.reduce_accum(frame, dst=0)
        frames.append(frame.numpy().copy())
    if rank == 0:
        np.save(out_path, np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


def test_repeated_steps_reduce_to_the_same_frame(tmp_path):
    import torch.multiprocessing as mp
    from helpers import oracle_render
    from rayzen_amd import scene as S
    out = str(tmp_path / "frames.npy")
    mp.spawn(_bench_like_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    frames = np.load(out)
    want = oracle_render(S.cornell_scene(), 40, 24, 2, 2, nthreads=2)
    for f in frames:
        assert (f.view(np.uint32) == want.view(np.uint32)).all()


# ---- bench.py --gpus N started as ONE plain process -----------------------------------------------

def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rz_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_mode_of_the_bench():
    b = _bench()
    assert b.launch_mode(1, {}) == "single"
    assert b.launch_mode(8, {}) == "local-group"                        # `python3 bench.py --gpus 8`: no launcher, no re-exec
    assert b.launch_mode(8, {"WORLD_SIZE": "8", "RANK": "3"}) == "ranks"  # torch.distributed.run: one process per GPU
    assert b.launch_mode(1, {"WORLD_SIZE": "1"}) == "single"
    with pytest.raises(SystemExit):
        b.launch_mode(4, {"WORLD_SIZE": "8"})


def test_bench_with_gpus_2_and_no_launcher_reaches_the_one_process_group(monkeypatch):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset used to exit with "launch N > 1 with torch.distributed.run".  It now
    takes the library's one-process group: rz_group_create(2) (ncclCommInitAll) -- no torch.distributed, no second process.
    There is no GPU here, so the device count and the group constructor are stand-ins: the test asserts that main() gets as
    far as Group.create(2, ...) and that no process group was initialised on the way."""
    import torch.distributed as dist
    from rayzen_amd import _lib, dist as rzdist
    b = _bench()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)

    class Reached(Exception):
        pass

    calls = []

    def fake_create(cls, ndev, devices=None, flags=0):
        calls.append((ndev, devices, flags))
        if len(calls) == 1:
            raise RuntimeError("ncclCommInitAll(2): unhandled system error (a stand-in)")      # the RCCL group cannot be made ...
        raise Reached()

    monkeypatch.setattr(_lib.hip(), "rz_device_count", lambda: 2)
    monkeypatch.setattr(rzdist.Group, "create", classmethod(fake_create))
    with pytest.raises(Reached):
        b.main(["--gpus", "2", "--steps", "1", "--warmup", "0", "--mesh-n", "4", "--width", "64", "--height", "32", "--no-cpu-baseline"])
    # first the RCCL group (devices None: 0..N-1, ncclCommInitAll); when THAT cannot be made the run does not die -- round 5 --
    # but falls back, in the same process, to a group without a communicator whose exchange step is device copies between the GPUs
    assert calls[0] == (2, None, 0)
    assert calls[1] == (2, [0, 1], rzdist.GROUP_LOOPBACK)
    assert not dist.is_initialized()
    # fewer devices than ranks: a clear refusal, not a hang
    monkeypatch.setattr(_lib.hip(), "rz_device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        b.main(["--gpus", "2", "--no-cpu-baseline"])
    assert "only 1 HIP device" in str(e.value)


# ---- the tile gather (round 4: rz_group_reduce's default transport), rehearsed on the CPU -------------------------------

@pytest.mark.parametrize("size,nranks", [((72, 40), 2), ((100, 45), 3), ((161, 67), 8), ((8, 8), 5)])
def test_pack_and_unpack_tiles_round_trip(size, nranks):
    """Every pixel of the frame comes back from its owner's packed set; a rank's set holds exactly its own tiles."""
    W, H = size
    rng = np.random.default_rng(7)
    frame = rng.random((H, W, 4), dtype=np.float32)
    own = D.owner_map(W, H, nranks)
    sets = []
    for r in range(nranks):
        mine = np.where((own == r)[..., None], frame, np.float32(0))     # what rank r's accumulation buffer holds
        s = D.pack_tiles(mine, r, nranks)
        assert s.shape == (D.tiles_per_rank(W, H, nranks), 64, 4)
        assert np.count_nonzero(s) == np.count_nonzero(mine)
        sets.append(s)
    got = D.unpack_tiles(np.stack(sets), W, H)
    assert (got.view(np.uint32) == frame.view(np.uint32)).all()


def _gather_worker(rank, world, port, W, H, spp, bounces, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from oracle import rzo
    from rayzen_amd import scene as S
    from helpers import oracle_frame, oracle_scene
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = S.cornell_scene()
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, spp, bounces)
    tx, ty = D.tile_grid(W, H)
    acc = np.zeros((H, W, 4), np.float32)
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        rzo.render(osc, fr, accum=acc, crop=(x0, y0, min(x0 + 8, W), min(y0 + 8, H)), nthreads=1)
    frame = D.gather_tiles(acc, rank, world, dst=1)          # (a root other than 0)
    if rank == 1:
        np.save(out_path, frame)
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharded_render_plus_tile_gather_is_bit_identical_to_single_rank(tmp_path):
    """world_size 2 over gloo: each rank renders its own tiles (the oracle stands in for the GPU), packs them, rank 1 gathers and
    scatters -- the data flow of rz_group_reduce's default transport -- and holds the single-rank frame bit for bit."""
    import torch.multiprocessing as mp
    from helpers import oracle_render
    from rayzen_amd import scene as S
    W, H, spp, bounces = 75, 41, 2, 3
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_gather_worker, args=(2, _free_port(), W, H, spp, bounces, out), nprocs=2, join=True)
    got = np.load(out)
    want = oracle_render(S.cornell_scene(), W, H, spp, bounces, nthreads=2)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
