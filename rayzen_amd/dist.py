"""Multi-GPU: pixel tiles sharded across ranks, one reduce on the accumulation buffer.

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


def reduce_accum(tensor, dst=0, group=None):
    """Sum the per-rank full-frame accumulation buffers onto rank `dst` (in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return tensor
