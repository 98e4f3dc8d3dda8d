"""The multi-GPU group of the C-ABI (rz_group_*, include/rayzen_hip.h).  CPU part: RCCL is bound at run time and every
entry point the group needs resolves; arguments are validated before any device is touched.  GPU part (one GPU on the
box): a 1-rank group goes through the real ncclCommInitRank / ncclCommInitAll + ncclReduce and must hand back the
single-context frame bit for bit."""
import numpy as np
import pytest

from rayzen_amd import dist as D
from rayzen_amd import _lib


def test_rccl_binds_at_run_time_without_a_gpu():
    v = D.rccl_version()
    assert v >= 21800, v            # RCCL of ROCm 7.x reports an NCCL 2.2x version number


def test_library_has_no_link_time_dependency_on_rccl():
    import subprocess
    out = subprocess.run(["readelf", "-d", _lib.HIP_SO], capture_output=True, text=True).stdout
    needed = [l for l in out.splitlines() if "NEEDED" in l]
    assert needed and not any("rccl" in l for l in needed), needed


def test_group_arguments_are_validated_before_any_device_is_touched():
    L = _lib.hip()
    assert not L.rz_group_create(0, None, 0)
    assert b"ndev" in L.rz_group_last_error(None)
    uid = bytes(128)
    import ctypes as C
    buf = (C.c_char * 128).from_buffer_copy(uid)
    assert not L.rz_group_create_rank(0, 3, 2, buf, 0)
    assert b"rank 3 of 2" in L.rz_group_last_error(None)
    assert not L.rz_group_create_rank(0, 0, 1, None, 0)
    assert L.rz_group_size(None) == 0 and L.rz_group_local_count(None) == 0 and L.rz_group_rank(None, 0) == -1
    assert L.rz_group_render(None) == -1 and L.rz_group_reduce(None, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["rank", "all"])
def test_one_rank_group_renders_reduces_and_matches_the_oracle(how):
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import frame_params
    from helpers import hip_render, oracle_render
    sc = S.bunny_scene(n=8, extras=True)
    W, H, spp, b = 96, 54, 3, 4
    g = D.Group.create_rank(0, 0, 1, D.unique_id()) if how == "rank" else D.Group.create(1)
    assert g.size == 1 and g.local_count == 1 and g.rank(0) == 0
    g.upload_scene(sc)
    g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    for _ in range(2):                  # the reduce is out of place: a second frame is not polluted by the first
        g.render()
        g.reduce(0)
    got = g.read_frame()
    g.close()
    ref = oracle_render(sc, W, H, spp, b)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    assert (got.view(np.uint32) == hip_render(sc, W, H, spp, b).view(np.uint32)).all()


@pytest.mark.gpu
def test_group_frame_continues_with_sample_base():
    """Chunked accumulation through the group: members' accumulation buffers are never overwritten by the reduce."""
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import frame_params
    from helpers import oracle_render
    sc = S.bunny_scene(n=8, extras=True)
    W, H, b = 64, 40, 4
    g = D.Group.create(1)
    g.upload_scene(sc)
    for base, k in ((0, 2), (2, 3)):
        g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, k, base))
        g.render()
        g.reduce(0)
    got = g.read_frame()
    g.close()
    ref = oracle_render(sc, W, H, 5, b)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.gpu
def test_cpp_group_example_is_bit_identical_to_a_single_context(tmp_path):
    """examples/render_group.cpp: C++ frontend -> GroupRenderer (Renderer.h) -> rz_group_* -> RCCL; the program itself
    compares the reduced frame with a single-context render and exits 0 only if they are the same bytes."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "render_group")
    lib = os.path.join(root, "rayzen_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "include"), "-I",
                           os.path.join(root, "rayzen_amd", "csrc", "host"),
                           os.path.join(root, "examples", "render_group.cpp"), "-L", lib, "-lrayzen_host",
                           "-lrayzen_hip", f"-Wl,-rpath,{lib}", "-o", exe])
    out = subprocess.run([exe, "0", "160", "96", "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-identical" in out.stdout


@pytest.mark.gpu
def test_two_contexts_on_two_devices_interleaved():
    """a context may be driven while ANOTHER device is current (the members of a one-process group).  Every exported
    entry point binds its context's device first (guarded(), rz_context.hip); without that, resolve / present / counters of a
    member allocate and launch on the wrong GPU.  Needs two devices: skipped on the one-GPU test box, run by whoever has a node."""
    L = _lib.hip()
    if L.rz_device_count() < 2:
        pytest.skip("needs two HIP devices")
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import Renderer, frame_params
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from helpers import oracle_render
    sc = S.bunny_scene(n=6, extras=True)
    W, H, spp, b = 64, 40, 4, 4
    ref = oracle_render(sc, W, H, spp, b)
    rs = [Renderer(0), Renderer(1)]
    for r in rs:
        r.upload_scene(sc)
        r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    rs[0].render(); rs[1].render_counted(); rs[0].resolve_rgba8(); rs[1].present(); rs[1].clear_accum(); rs[1].render()
    rs[0].clear_accum(); rs[0].render_counted()
    for r in rs:
        got = r.read_accum()
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()
        r.close()
    # the one-process group over both devices: tiles dealt to two ranks, one RCCL reduce
    g = D.Group.create(2)
    assert g.size == 2 and g.local_count == 2
    g.upload_scene(sc)
    g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    g.render(); g.reduce(0); g.sync()
    frame = g.read_frame()
    root_ms, max_ms = g.last_reduce_ms()
    g.close()
    assert (frame.view(np.uint32) == ref.view(np.uint32)).all()
    assert root_ms >= 0.0 and max_ms >= root_ms
    # ... the same through the tile gather (ncclSend / ncclRecv between the two devices), and through plain device copies
    for make in (lambda: D.Group.create(2), lambda: D.Group.create(2, devices=[0, 1], flags=D.GROUP_LOOPBACK)):
        g = make()
        if not g.transport.startswith("tile-gather"):
            g.set_transport("gather")
        g.upload_scene(sc)
        g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
        g.render(); g.reduce(1); g.sync()
        frame = g.read_frame()
        how = g.transport
        g.close()
        assert (frame.view(np.uint32) == ref.view(np.uint32)).all(), how


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,root,size", [(2, 0, (96, 54)), (3, 2, (100, 45)), (8, 5, (161, 67)), (5, 0, (8, 8))])
def test_loopback_group_of_n_ranks_gathers_the_single_gpu_frame(nranks, root, size):
    """The tile gather of an N-rank group, rehearsed on one GPU (RZ_GROUP_LOOPBACK: N members on device 0, device copies in
    place of ncclSend / ncclRecv): the dealing of tiles, each member's packing, the root's scatter and the stream ordering
    are the code an N-GPU group runs.  Frame sizes that are no multiple of the 8 x 8 tile, fewer tiles than ranks, a root
    other than 0, and a second frame continued with sample_base: bit-identical to the oracle and to one context."""
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import frame_params
    from helpers import hip_render, oracle_render
    sc = S.bunny_scene(n=8, extras=True)
    (W, H), b = size, 4
    g = D.Group.create(nranks, devices=[0] * nranks, flags=D.GROUP_LOOPBACK)
    assert g.size == nranks and g.local_count == nranks and g.transport.startswith("tile-gather(loopback")
    g.upload_scene(sc)
    for base, k in ((0, 2), (2, 3)):
        g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, k, base))
        g.render()
        g.reduce(root)
    got = g.read_frame()
    root_ms, max_ms = g.last_reduce_ms()
    g.close()
    ref = oracle_render(sc, W, H, 5, b)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    assert (got.view(np.uint32) == hip_render(sc, W, H, 5, b).view(np.uint32)).all()
    assert root_ms >= 0.0 and max_ms >= root_ms


@pytest.mark.gpu
def test_one_rank_group_transports(monkeypatch):
    """The default exchange step is north_star's ONE ncclReduce(sum) of the whole buffers (round 5; round 4 had made the never-run
    tile gather the default).  RZ_GROUP_TRANSPORT=gather at creation, or rz_group_set_transport at any time, selects the gather;
    the frame is the same bits either way, also when the transport changes between two chunks of one frame."""
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import frame_params
    from helpers import oracle_render
    sc = S.bunny_scene(n=8, extras=True)
    W, H, b = 96, 54, 4
    ref = oracle_render(sc, W, H, 5, b)
    monkeypatch.delenv("RZ_GROUP_TRANSPORT", raising=False)
    g = D.Group.create(1)
    assert g.transport == "rccl-reduce"
    g.upload_scene(sc)
    g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, 2, 0))
    g.render(); g.reduce(0)
    g.set_transport("gather")
    assert g.transport == "tile-gather(rccl send/recv)"
    g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, 3, 2))
    g.render(); g.reduce(0)
    got = g.read_frame()
    with pytest.raises(Exception):
        g.set_transport("carrier pigeon")
    g.set_transport("reduce")
    assert g.transport == "rccl-reduce"
    g.close()
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    monkeypatch.setenv("RZ_GROUP_TRANSPORT", "gather")
    g = D.Group.create(1)
    assert g.transport == "tile-gather(rccl send/recv)"
    g.close()


def test_loopback_is_refused_where_it_makes_no_sense():
    L = _lib.hip()
    import ctypes as C
    buf = (C.c_char * 128).from_buffer_copy(bytes(128))
    assert not L.rz_group_create_rank(0, 0, 1, buf, D.GROUP_LOOPBACK)
    assert b"RZ_GROUP_LOOPBACK" in L.rz_group_last_error(None)
    assert L.rz_group_transport(None) == b""
    assert L.rz_group_set_transport(None, b"reduce") != 0


@pytest.mark.gpu
def test_group_beside_torch_distributed_in_one_process(tmp_path):
    """What bench.py's launcher mode does, for one rank: torch FIRST (its wheel bundles a HIP runtime and an RCCL of its own;
    a process that loads librayzen_hip.so before torch leaves torch without a GPU -- profiles/scripts/hip_runtime_order.py), an
    `nccl` process group, THEN the library, which must bind to the runtime and the RCCL already in the process, form its own
    communicator from an id broadcast over torch's, render, land the frame, and agree with the oracle.  In a child process:
    this one has the library loaded already."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "beside_torch.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import numpy as np
from rayzen_amd import dist as D, scene as S
from rayzen_amd.renderer import frame_params
from helpers import oracle_render
uid = torch.zeros(128, dtype=torch.uint8, device=dev)
uid.copy_(torch.frombuffer(bytearray(D.unique_id()), dtype=torch.uint8))
dist.broadcast(uid, src=0)
g = D.Group.create_rank(0, 0, 1, bytes(uid.cpu().numpy().tobytes()))
sc = S.bunny_scene(n=8, extras=True)
W, H, spp, b = 96, 54, 3, 4
g.upload_scene(sc); g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
g.render(); g.reduce(0); dist.barrier(); g.sync()
got = g.read_frame(); g.close()
maps = sorted({{l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "librccl" in l}})
assert len([m for m in maps if "amdhip" in m]) == 1 and len([m for m in maps if "librccl" in m]) == 1, maps
assert (got.view(np.uint32) == oracle_render(sc, W, H, spp, b).view(np.uint32)).all()
dist.destroy_process_group()
print("OK one HIP runtime, one RCCL:", maps)
''')
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK one HIP runtime" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
