"""The TLAS walk of rz_trace.h keeps no stack: it follows the list of nodes in the order the shader's loop pops them
(FS:464-501, right child first), with skip positions.  These scenes give it TLAS shapes RayZen's own builder
(BVH.cpp:178-240: one instance per leaf, balanced) never produces -- the C-ABI takes any node array -- and compare with
the oracle's literal stack loop, bit for bit, tallies included."""
import numpy as np
import pytest

from helpers import hip_render, oracle_render
from rayzen_amd import scene as S

pytestmark = pytest.mark.gpu


def _eq(g, o):
    assert g.shape == o.shape
    assert (g.view(np.uint32) == o.view(np.uint32)).all()


def _cubes(count):
    sc = S.Scene()
    cube = sc.add_mesh(S.make_cube(0))
    mirror = sc.add_mesh(S.make_cube(2))
    for i in range(count):
        x, y = (i % 5) * 1.6 - 3.2, (i // 5) * 1.5 - 1.2
        sc.add_object(mirror if i % 3 == 0 else cube, S.scale(S.translate(S.identity(), (x, y, -5.0 - 0.3 * (i % 4))), (0.5, 0.5, 0.5)))
    sc.build()
    return sc


def _world_boxes(sc):
    """instance -> (min, max) of its world box, read off the leaves RayZen's builder made."""
    nodes, idx = sc.arrays[S.BIND_TLAS_NODES], sc.arrays[S.BIND_TLAS_INDICES]
    out = {}
    for n in nodes:
        if n["count"] > 0:
            for k in range(n["count"]):
                out[int(idx[n["leftFirst"] + k])] = (n["boundsMin"].copy(), n["boundsMax"].copy())
    return out


def _union(boxes):
    return np.min([b[0] for b in boxes], axis=0), np.max([b[1] for b in boxes], axis=0)


def _set_tlas(sc, nodes, idx):
    sc.arrays[S.BIND_TLAS_NODES] = np.array(nodes, S.BVH_NODE)
    sc.arrays[S.BIND_TLAS_INDICES] = np.array(idx, np.int32)


def _node(box, left_first, count):
    return (box[0], left_first, box[1], count)


def _check(sc, w=96, h=54, spp=4, bounces=4):
    g, gc = hip_render(sc, w, h, spp, bounces, counted=True)
    o, oc = oracle_render(sc, w, h, spp, bounces, want_counters=True)
    _eq(g, o)
    for k in ("tlas_nodes", "tlas_leaf_indices", "instances", "blas_nodes", "triangles"):
        assert gc[k] == oc[k], k
    return g


def test_leaves_holding_several_instances():
    sc = _cubes(10)
    wb = _world_boxes(sc)
    groups = [[0, 1, 2, 3], [4], [5, 6], [7, 8, 9]]
    idx = [i for g in groups for i in g]
    starts = np.cumsum([0] + [len(g) for g in groups])
    gb = [_union([wb[i] for i in g]) for g in groups]
    # root -> (1: internal over groups 0,1   2: internal over groups 2,3) -> leaves 3,4 and 5,6
    nodes = [_node(_union(gb), 1, -1), _node(_union(gb[:2]), 3, -1), _node(_union(gb[2:]), 5, -1),
             _node(gb[0], starts[0], 4), _node(gb[1], starts[1], 1), _node(gb[2], starts[2], 2), _node(gb[3], starts[3], 3)]
    _set_tlas(sc, nodes, idx)
    _check(sc)
    _eq(hip_render(sc, 96, 54, 64, 3), oracle_render(sc, 96, 54, 64, 3))           # the compacting persistent launch is not taken at this size, the 64-lane mapping is


def test_a_chain_deeper_than_the_shaders_stack():
    """A right-leaning chain of 70 links (left child a leaf, right child the next link): the left leaf of every link
    waits on the stack while the walk goes down the right side, so from link 63 on the shader's stack[64] would
    overflow; the oracle drops those pushes (oracle/rz_oracle.c), so nothing below link 62 is ever visited and the
    instance that hangs only at the bottom stays invisible."""
    sc = _cubes(8)
    reference = hip_render(sc, 96, 54, 4, 4)                   # RayZen's own TLAS: all eight cubes
    wb = _world_boxes(sc)
    allb = _union(list(wb.values()))
    links = 70
    nodes, idx = [None] * (2 * links + 1), []
    for k in range(links):                                      # link k at 2k, its children at 2k+1 (leaf) and 2k+2
        nodes[2 * k] = _node(allb, 2 * k + 1, -1)
        nodes[2 * k + 1] = _node(wb[k % 7], len(idx), 1)        # instances 0..6 hang off the chain again and again
        idx.append(k % 7)
    nodes[2 * links] = _node(wb[7], len(idx), 1)                # instance 7 only at the very bottom
    idx.append(7)
    _set_tlas(sc, nodes, idx)
    g = _check(sc)
    assert (g.view(np.uint32) != reference.view(np.uint32)).any()       # the cut-off is visible in this frame


def test_left_leaning_chain_is_walked_to_the_bottom():
    """The mirror image (left child the next link, right child a leaf): the stack never holds more than one waiting
    entry, all 70 levels are walked, and the frame is the one RayZen's own TLAS gives."""
    sc = _cubes(8)
    reference = hip_render(sc, 96, 54, 4, 4)
    wb = _world_boxes(sc)
    allb = _union(list(wb.values()))
    links = 70
    nodes, idx = [None] * (2 * links + 1), []
    for k in range(links):                                      # link k at (0 if k == 0 else 2k-1), children at 2k+1 (next link), 2k+2 (leaf)
        nodes[0 if k == 0 else 2 * k - 1] = _node(allb, 2 * k + 1, -1)
        nodes[2 * k + 2] = _node(wb[k % 7], len(idx), 1)
        idx.append(k % 7)
    nodes[2 * links - 1] = _node(wb[7], len(idx), 1)            # instance 7 only at the very bottom
    idx.append(7)
    _set_tlas(sc, nodes, idx)
    g = _check(sc)
    # every cube is found; a cube reached through several leaves is intersected several times, which changes no closest hit
    _eq(g, reference)
