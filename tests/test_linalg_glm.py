"""Instance fill is GLM-faithful (SURVEY.md section 8a row "flatten / instance fill").

RayZen fills BVHInstance::inverseTransform with glm::inverse (RayZen/src/main.cpp:1001, 1058, 1151) and the world box of
an instance from transform * vec4(corner, 1) (main.cpp:974-993, 1168-1191).  GLM is a third-party, un-vendored,
version-unpinned dependency of the reference (RayZen/CMakeLists.txt:16-18; install_requirements.sh installs the distro's
libglm-dev = GLM 0.9.9.8) and is absent from this image, so the host library (rz_linalg.h), the oracle
(rz_oracle_bvh.c) and the device TLAS builder (rz_tlas_device.hip; GPU test in test_gpu_cases.py) each restate GLM's
published algorithm.  Pinned here by hand-derived known answers and by a third, numpy-float32 statement of the same
sequence of roundings (every product, difference and sum its own binary32 operation)."""
import numpy as np
import pytest

from oracle import rzo
from rayzen_amd import _lib
from rayzen_amd import scene as S

f32 = np.float32


def host_inverse(m):
    return S.inverse(np.asarray(m, np.float32))


def cm(rows):
    """16 floats, column-major, from a 4x4 given row by row (as one writes a matrix on paper)."""
    return np.asarray(rows, np.float32).T.reshape(16).copy()


def glm_inverse_numpy(m):
    """compute_inverse<4, 4> of GLM 0.9.9.8 (glm/detail/func_matrix.inl), one np.float32 operation per GLM operation."""
    M = lambda c, r: f32(m[4 * c + r])
    d = lambda a, b, c, e: f32(f32(a * b) - f32(c * e))
    C = {}
    C[0] = d(M(2, 2), M(3, 3), M(3, 2), M(2, 3)); C[2] = d(M(1, 2), M(3, 3), M(3, 2), M(1, 3)); C[3] = d(M(1, 2), M(2, 3), M(2, 2), M(1, 3))
    C[4] = d(M(2, 1), M(3, 3), M(3, 1), M(2, 3)); C[6] = d(M(1, 1), M(3, 3), M(3, 1), M(1, 3)); C[7] = d(M(1, 1), M(2, 3), M(2, 1), M(1, 3))
    C[8] = d(M(2, 1), M(3, 2), M(3, 1), M(2, 2)); C[10] = d(M(1, 1), M(3, 2), M(3, 1), M(1, 2)); C[11] = d(M(1, 1), M(2, 2), M(2, 1), M(1, 2))
    C[12] = d(M(2, 0), M(3, 3), M(3, 0), M(2, 3)); C[14] = d(M(1, 0), M(3, 3), M(3, 0), M(1, 3)); C[15] = d(M(1, 0), M(2, 3), M(2, 0), M(1, 3))
    C[16] = d(M(2, 0), M(3, 2), M(3, 0), M(2, 2)); C[18] = d(M(1, 0), M(3, 2), M(3, 0), M(1, 2)); C[19] = d(M(1, 0), M(2, 2), M(2, 0), M(1, 2))
    C[20] = d(M(2, 0), M(3, 1), M(3, 0), M(2, 1)); C[22] = d(M(1, 0), M(3, 1), M(3, 0), M(1, 1)); C[23] = d(M(1, 0), M(2, 1), M(2, 0), M(1, 1))
    Fac = [np.array([C[b], C[b], C[b + 2], C[b + 3]], f32) for b in (0, 4, 8, 12, 16, 20)]
    Vec = [np.array([M(1, k), M(0, k), M(0, k), M(0, k)], f32) for k in range(4)]
    with np.errstate(all="ignore"):
        Inv0 = (Vec[1] * Fac[0] - Vec[2] * Fac[1]) + Vec[3] * Fac[2]
        Inv1 = (Vec[0] * Fac[0] - Vec[2] * Fac[3]) + Vec[3] * Fac[4]
        Inv2 = (Vec[0] * Fac[1] - Vec[1] * Fac[3]) + Vec[3] * Fac[5]
        Inv3 = (Vec[0] * Fac[2] - Vec[1] * Fac[4]) + Vec[2] * Fac[5]
        SignA, SignB = np.array([1, -1, 1, -1], f32), np.array([-1, 1, -1, 1], f32)
        cols = [Inv0 * SignA, Inv1 * SignB, Inv2 * SignA, Inv3 * SignB]
        Row0 = np.array([cols[0][0], cols[1][0], cols[2][0], cols[3][0]], f32)
        Dot0 = np.array([M(0, 0), M(0, 1), M(0, 2), M(0, 3)], f32) * Row0
        Dot1 = f32(f32(Dot0[0] + Dot0[1]) + f32(Dot0[2] + Dot0[3]))
        one_over = f32(1.0) / Dot1
        return np.concatenate([c * one_over for c in cols]).astype(f32)


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


IMPLS = [("host rz_linalg.h", host_inverse), ("oracle", rzo.mat4_inverse), ("numpy statement", glm_inverse_numpy)]


@pytest.mark.parametrize("name,inv", IMPLS)
def test_inverse_known_answers(name, inv):
    # (1) translate(1, 2, 3) . scale(2, 4, 8): every cofactor an exact integer, det = 64 -> exact result, derived by hand:
    #     inverse = scale(1/2, 1/4, 1/8) . translate(-1, -2, -3)
    m = cm([[2, 0, 0, 1], [0, 4, 0, 2], [0, 0, 8, 3], [0, 0, 0, 1]])
    want = cm([[0.5, 0, 0, -0.5], [0, 0.25, 0, -0.5], [0, 0, 0.125, -0.375], [0, 0, 0, 1]])
    assert (inv(m) == want).all(), name             # (== : a cofactor that is -0 where `want` says 0 is still right)
    # (2) a quarter turn about +y, x -> -z, z -> +x, translated by (5, 6, 7): orthonormal, det = 1 -> inverse = transpose
    #     of the rotation and the translation -(R^T t) = (7, -6, -5)
    m = cm([[0, 0, 1, 5], [0, 1, 0, 6], [-1, 0, 0, 7], [0, 0, 0, 1]])
    want = cm([[0, 0, -1, 7], [0, 1, 0, -6], [1, 0, 0, -5], [0, 0, 0, 1]])
    assert (inv(m) == want).all(), name
    # (3) singular, diag(1, 1, 1, 0): GLM neither tests nor throws.  By hand: only Coef11 = m11 m22 - m21 m12 = 1 is
    #     nonzero, so the cofactor matrix is zero but for [3][3] = 1; Row0 = (0, -0, 0, -0), Dot1 = +0, 1 / Dot1 = +inf;
    #     0 * inf = NaN everywhere, 1 * inf = +inf at [3][3]
    got = inv(cm([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 0]]))
    assert np.isnan(got[:15]).all() and got[15] == np.inf, (name, got)
    # (4) the zero matrix: every entry 0 * inf = NaN
    assert np.isnan(inv(np.zeros(16, np.float32))).all(), name


def test_inverse_three_statements_agree_bit_for_bit():
    rng = np.random.default_rng(7)
    mats = []
    for k in range(400):
        m = rng.normal(size=16).astype(np.float32) * np.float32(10.0 ** rng.integers(-3, 4))
        if k % 3 == 0:                      # affine, like every GameObject::transform
            m[3], m[7], m[11], m[15] = 0, 0, 0, 1
        mats.append(m)
    # RayZen's own transforms (main.cpp:378-384) and C4's rotating instances
    I = S.identity()
    mats += [S.translate(S.scale(I, (8.0, 0.5, 8.0)), (0.0, -3.0, 0.0)), S.translate(S.scale(I, (1.2, 1.2, 1.2)), (2.5, 0.8, 2.5))]
    mats += S.instanced_transforms(7, 16)
    for m in mats:
        a, b, c = host_inverse(m), rzo.mat4_inverse(m), glm_inverse_numpy(m)
        assert same_bits(a, b) and same_bits(a, c), m


def test_inverse_differs_from_the_round_2_formula_somewhere():
    """The evaluation order is observable: an adjugate summed in another order gives other bits for some matrices (round
    2's rz_linalg.h was such a formula).  Guards against 'fixing' the order back."""
    rng = np.random.default_rng(11)
    differs = 0
    for _ in range(200):
        m = rng.normal(size=16).astype(np.float32)
        exact = np.linalg.inv(m.astype(np.float64).reshape(4, 4).T).T.reshape(16).astype(np.float32)
        differs += not same_bits(host_inverse(m), exact)
    assert differs > 100          # correctly rounded inverses are NOT what GLM returns: its roundings are part of the contract


def _world_box(fn, root, m):
    return fn(root, m)


def test_world_box_adds_the_column_products_pairwise():
    """glm mat4 * vec4 = (m[0] x + m[1] y) + (m[2] z + m[3] w).  Row (1e8, 1, -1e8, 1) on the corner (1, 1, 1): pairwise
    (1e8 + 1) + (-1e8 + 1) = 1e8 + -1e8 = 0 in binary32; left to right it would be ((1e8 + 1) + -1e8) + 1 = 1."""
    m = cm([[1e8, 1, -1e8, 1], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    root = np.zeros(1, S.BVH_NODE)[0]
    root["boundsMin"], root["boundsMax"] = (1, 1, 1), (1, 1, 1)          # a degenerate box: all 8 corners are (1, 1, 1)
    mn_o, mx_o = rzo.world_bounds(root, m)
    mn_h, mx_h = np.zeros(3, np.float32), np.zeros(3, np.float32)
    r1 = np.zeros(1, S.BVH_NODE)
    r1[0] = root
    _lib.host().rzh_world_bounds(r1.ctypes.data, m.ctypes.data, mn_h.ctypes.data, mx_h.ctypes.data)
    for mn, mx in ((mn_o, mx_o), (mn_h, mx_h)):
        assert mn[0] == 0.0 and mx[0] == 0.0 and mn[1] == 1.0 and mn[2] == 1.0, (mn, mx)


def test_world_boxes_of_host_and_oracle_agree_on_random_instances():
    rng = np.random.default_rng(3)
    for _ in range(200):
        m = S.rotate(S.translate(S.identity(), rng.normal(size=3) * 5), float(rng.normal()), rng.normal(size=3))
        m = S.scale(m, np.abs(rng.normal(size=3)) + 0.1)
        root = np.zeros(1, S.BVH_NODE)
        lo = rng.normal(size=3).astype(np.float32)
        root[0]["boundsMin"], root[0]["boundsMax"] = lo, lo + np.abs(rng.normal(size=3)).astype(np.float32)
        mn_o, mx_o = rzo.world_bounds(root[0], m)
        mn_h, mx_h = np.zeros(3, np.float32), np.zeros(3, np.float32)
        _lib.host().rzh_world_bounds(root.ctypes.data, np.ascontiguousarray(m, np.float32).ctypes.data, mn_h.ctypes.data, mx_h.ctypes.data)
        assert same_bits(mn_o, mn_h) and same_bits(mx_o, mx_h)
