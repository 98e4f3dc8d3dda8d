"""Multi-GPU: pixel tiles sharded across ranks, one exchange step on the accumulation buffers (a gather of each rank's own
tiles -- the C-ABI group's default since round 4, mirrored here by pack_tiles / unpack_tiles / gather_tiles -- or a reduce).

RayZen has no multi-GPU code.  Pixels are independent (fragment_shader.glsl:668-773
touches only its own gl_FragCoord), so the frame is cut into 8x8-pixel tiles
(one wavefront each) dealt round-robin, tile t -> rank t % nranks, which
spreads the expensive centre of the image and the cheap sky evenly.  Every
rank renders ALL samples of its own pixels (the shader's currentIor carries
from sample to sample, so a pixel's samples cannot be split) into a
zero-initialised full-frame RGBA32F buffer; since the tile sets are disjoint,
`reduce(SUM)` adds each pixel's single real value to zeros and the result is
bit-identical to a single-GPU render.  One process per GPU; the collective is
torch.distributed's (backend "nccl" is RCCL over xGMI on ROCm; "gloo" for the
CPU rehearsal in tests).
"""
import numpy as np

TILE_W = 8
TILE_H = 8


def tile_grid(width, height):
    return (width + TILE_W - 1) // TILE_W, (height + TILE_H - 1) // TILE_H


def owner_map(width, height, nranks):
    """(H, W) int32: the rank that owns each pixel."""
    tx, ty = tile_grid(width, height)
    tiles = (np.arange(tx * ty, dtype=np.int64) % nranks).astype(np.int32).reshape(ty, tx)
    return np.repeat(np.repeat(tiles, TILE_H, axis=0), TILE_W, axis=1)[:height, :width]


def local_tile_count(width, height, rank, nranks):
    tx, ty = tile_grid(width, height)
    return (tx * ty - rank + nranks - 1) // nranks


def owned_samples(width, height, spp, rank, nranks):
    """Number of camera paths (pixels x spp) rank `rank` renders."""
    return int((owner_map(width, height, nranks) == rank).sum()) * spp


def tiles_per_rank(width, height, nranks):
    """Tiles in one rank's packed set (the same for every rank: the last ranks' sets end in zero tiles)."""
    tx, ty = tile_grid(width, height)
    return (tx * ty + nranks - 1) // nranks


def pack_tiles(accum, rank, nranks):
    """The host mirror of rz_pack_tiles (rayzen_amd/csrc/hip/rz_group.hip): rank `rank`'s own tiles of the (H, W, 4) buffer in the
    order of its local tiles (local tile lt is tile lt * nranks + rank), 64 pixels each, pixel l of a tile at (l & 7, l >> 3);
    pixels beyond the frame's edge and tiles beyond the last one are zeros.  -> (tiles_per_rank, 64, 4) float32."""
    h, w = accum.shape[:2]
    tx, ty = tile_grid(w, h)
    padded = np.zeros((ty * TILE_H, tx * TILE_W, 4), np.float32)
    padded[:h, :w] = accum
    tiles = padded.reshape(ty, TILE_H, tx, TILE_W, 4).transpose(0, 2, 1, 3, 4).reshape(tx * ty, TILE_H * TILE_W, 4)
    out = np.zeros((tiles_per_rank(w, h, nranks), TILE_H * TILE_W, 4), np.float32)
    mine = tiles[rank::nranks]
    out[:mine.shape[0]] = mine
    return out


def unpack_tiles(sets, width, height):
    """The host mirror of rz_unpack_tiles: the frame from the ranks' packed sets, (nranks, tiles_per_rank, 64, 4) -> (H, W, 4)."""
    nranks = sets.shape[0]
    tx, ty = tile_grid(width, height)
    tiles = np.zeros((tx * ty, TILE_H * TILE_W, 4), np.float32)
    for r in range(nranks):
        n = len(range(r, tx * ty, nranks))
        tiles[r::nranks] = sets[r][:n]
    padded = tiles.reshape(ty, tx, TILE_H, TILE_W, 4).transpose(0, 2, 1, 3, 4).reshape(ty * TILE_H, tx * TILE_W, 4)
    return np.ascontiguousarray(padded[:height, :width])


def gather_tiles(accum, rank, nranks, dst=0, group=None):
    """The group's exchange step over torch.distributed (the CPU rehearsal of rz_group_reduce's default transport): every rank
    packs its own tiles, rank `dst` gathers the sets and scatters them into the frame.  Returns the frame on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(pack_tiles(accum, rank, nranks))
    if rank == dst:
        sets = [torch.empty_like(mine) for _ in range(nranks)]
        dist.gather(mine, gather_list=sets, dst=dst, group=group)
        return unpack_tiles(np.stack([s.numpy() for s in sets]), accum.shape[1], accum.shape[0])
    dist.gather(mine, gather_list=None, dst=dst, group=group)
    return None


def reduce_accum(tensor, dst=0, group=None):
    """Sum the per-rank full-frame accumulation buffers onto rank `dst` (in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return tensor


# ---------------------------------------------------------------------------------------------------------
# The multi-GPU group of the C-ABI (include/rayzen_hip.h: rz_group_*; rayzen_amd/csrc/hip/rz_group.hip): contexts +
# RCCL communicator owned by librayzen_hip.so.  This class is only a ctypes view of it.
# ---------------------------------------------------------------------------------------------------------

def rccl_version():
    """Version of the RCCL the library binds (no GPU needed)."""
    import ctypes as C
    from . import _lib
    v = C.c_int(0)
    L = _lib.hip()
    rc = L.rz_group_rccl_version(C.byref(v))
    if rc != 0:
        raise RuntimeError(f"rz_group_rccl_version failed ({rc}): {L.rz_group_last_error(None).decode()}")
    return v.value


def unique_id():
    """128-byte communicator id (ncclGetUniqueId): rank 0 makes it, every rank passes it to Group.create_rank."""
    import ctypes as C
    from . import _lib
    buf = (C.c_char * 128)()
    L = _lib.hip()
    rc = L.rz_group_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"rz_group_unique_id failed ({rc}): {L.rz_group_last_error(None).decode()}")
    return bytes(buf)


GROUP_LOOPBACK = 0x10000        # RZ_GROUP_LOOPBACK (include/rayzen_hip.h): N rehearsal ranks on the devices named, no communicator


class Group:
    """N tile-sharded contexts and their RCCL communicator(s)."""

    def __init__(self, handle):
        from . import _lib
        self._L = _lib.hip()
        self._g = handle
        self.width = self.height = 0

    @classmethod
    def create(cls, ndev, devices=None, flags=0):
        """One process driving ndev devices (ncclCommInitAll)."""
        import ctypes as C
        from . import _lib
        L = _lib.hip()
        arr = (C.c_int * ndev)(*devices) if devices is not None else None
        g = L.rz_group_create(int(ndev), arr, int(flags))
        if not g:
            raise RuntimeError("rz_group_create failed: " + L.rz_group_last_error(None).decode())
        return cls(g)

    @classmethod
    def create_rank(cls, device, rank, nranks, uid, flags=0):
        """One process per GPU (ncclCommInitRank); blocks until every rank has joined."""
        import ctypes as C
        from . import _lib
        L = _lib.hip()
        buf = (C.c_char * 128).from_buffer_copy(uid)
        g = L.rz_group_create_rank(int(device), int(rank), int(nranks), buf, int(flags))
        if not g:
            raise RuntimeError("rz_group_create_rank failed: " + L.rz_group_last_error(None).decode())
        return cls(g)

    def close(self):
        g, self._g = getattr(self, "_g", None), None
        if g:
            self._L.rz_group_destroy(g)

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            from .renderer import RayZenError
            raise RayZenError(what, rc, self._L.rz_group_last_error(self._g).decode())

    @property
    def size(self):
        return self._L.rz_group_size(self._g)

    @property
    def transport(self):
        """How reduce() moves the frame: 'rccl-reduce', 'rccl-reduce(fallback: ...)' or 'tile-gather(...)' (rz_group_transport)."""
        fn = getattr(self._L, "rz_group_transport", None)
        return fn(self._g).decode() if fn is not None and fn.restype is not None and fn.argtypes else "unknown (library predates rz_group_transport)"

    def set_transport(self, name):
        """'reduce' or 'gather' from the next reduce() on; every rank of the group must make the same call."""
        self._check(self._L.rz_group_set_transport(self._g, name.encode()), "rz_group_set_transport")

    @property
    def local_count(self):
        return self._L.rz_group_local_count(self._g)

    def rank(self, local=0):
        return self._L.rz_group_rank(self._g, local)

    def upload_scene(self, scene):
        from .scene import BINDING_DTYPES
        for b in BINDING_DTYPES:
            a = np.ascontiguousarray(scene.arrays[b])
            self._check(self._L.rz_group_upload(self._g, int(b), a.ctypes.data if a.nbytes else None, a.nbytes), "rz_group_upload")

    def update(self, binding, array, offset_bytes=0):
        a = np.ascontiguousarray(array)
        self._check(self._L.rz_group_update(self._g, int(binding), int(offset_bytes), a.ctypes.data if a.nbytes else None, a.nbytes),
                    "rz_group_update")

    def set_frame(self, params):
        import ctypes as C
        self._check(self._L.rz_group_set_frame(self._g, C.byref(params)), "rz_group_set_frame")
        self.width, self.height = params.width, params.height

    def render(self):
        self._check(self._L.rz_group_render(self._g), "rz_group_render")

    def reduce(self, root=0):
        self._check(self._L.rz_group_reduce(self._g, int(root)), "rz_group_reduce")

    def sync(self):
        self._check(self._L.rz_group_sync(self._g), "rz_group_sync")

    def read_frame(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._L.rz_group_read_frame(self._g, out.ctypes.data, out.nbytes), "rz_group_read_frame")
        return out

    def last_reduce_ms(self):
        """(ms on the root member or -1, longest ms over this process's members) of the last reduce (HIP events)."""
        import ctypes as C
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._L.rz_group_last_reduce_ms(self._g, C.byref(a), C.byref(b)), "rz_group_last_reduce_ms")
        return float(a.value), float(b.value)

    def frame_device_ptr(self):
        return self._L.rz_group_frame_device_ptr(self._g)

    def member(self, local=0):
        """A Renderer view of one member's context (per-device calls: counters, timings).  The group owns the context:
        closing or dropping the view does nothing."""
        from .renderer import Renderer

        class _MemberView(Renderer):
            def __init__(view, L, c, w, h):     # noqa: N805
                view._L, view._c, view.width, view.height = L, c, w, h

            def close(view):                    # noqa: N805
                view._c = None

            __del__ = close

        return _MemberView(self._L, self._L.rz_group_ctx(self._g, int(local)), self.width, self.height)
