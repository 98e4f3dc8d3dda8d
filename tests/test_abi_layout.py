"""The C-ABI libraries load, export every symbol include/*.h declares, and agree on struct layouts
(no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rayzen_amd import _lib
from rayzen_amd import scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, src)))


def test_hip_library_exports_every_declared_symbol():
    names = _declared("rayzen_hip.h", "rz_")
    assert len(names) >= 18
    lib = C.CDLL(_lib.HIP_SO)
    for n in names:
        assert hasattr(lib, n), f"librayzen_hip.so lacks {n}"
    assert set(names) == set(_lib.HIP_SYMBOLS)


def test_host_library_exports_every_declared_symbol():
    names = _declared("rayzen_host.h", "rzh_")
    lib = C.CDLL(_lib.HOST_SO)
    for n in names:
        assert hasattr(lib, n), f"librayzen_host.so lacks {n}"
    assert set(names) == set(_lib.HOST_SYMBOLS)


def test_struct_sizes_match_reference_ssbo_layouts():
    L = _lib.hip()
    # RayZen/include/Mesh.h:9-17 (64), BVH.h:7-12 (32), BVH.h:14-21 (144), Material.h (32), Light.h (32)
    assert [L.rz_sizeof(i) for i in range(5)] == [64, 32, 144, 32, 32]
    assert L.rz_sizeof(5) == C.sizeof(_lib.FrameParams)
    assert L.rz_sizeof(6) == C.sizeof(_lib.Counters) == 120     # 10 traversal-side tallies + 5 more of round 4 (the work model prices them)
    assert (S.TRIANGLE.itemsize, S.BVH_NODE.itemsize, S.BVH_INSTANCE.itemsize, S.MATERIAL.itemsize,
            S.LIGHT.itemsize) == (64, 32, 144, 32, 32)
    # field offsets of the SSBO structs (std430 == C++ layout)
    assert [S.TRIANGLE.fields[f][1] for f in ("v0", "v1", "v2", "materialIndex")] == [0, 16, 32, 48]
    assert [S.BVH_NODE.fields[f][1] for f in ("boundsMin", "leftFirst", "boundsMax", "count")] == [0, 12, 16, 28]
    assert [S.BVH_INSTANCE.fields[f][1] for f in ("blasNodeOffset", "blasTriOffset", "meshIndex", "globalTriOffset",
                                                  "transform", "inverseTransform")] == [0, 4, 8, 12, 16, 80]
    assert [S.MATERIAL.fields[f][1] for f in ("albedo", "metallic", "roughness", "reflectivity", "transparency",
                                              "ior")] == [0, 12, 16, 20, 24, 28]
    assert [S.LIGHT.fields[f][1] for f in ("positionOrDirection", "color", "power")] == [0, 16, 28]


def test_versions():
    assert b"gfx950" in _lib.hip().rz_version()
    assert _lib.host().rzh_version()


def test_library_carries_the_hash_of_the_sources_it_was_built_from():
    """the loaded library, the file on disk and the tree agree on one source hash (what bench.py checks
    before it prints another run's hardware counters)."""
    from rayzen_amd import build
    h = _lib.hip().rz_source_hash().decode()
    assert re.fullmatch(r"[0-9a-f]{64}", h), h
    assert build.stamped_hash(_lib.HIP_SO) == h
    if not os.environ.get("RAYZEN_HIP_SO"):
        assert h == build.source_hash()
    assert build.source_hash(("-DRZ_PROF",)) != build.source_hash()      # flags are part of the identity
    assert build.stamped_hash(os.path.join(ROOT, "no", "such.so")) is None


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device rz_create must fail; the Python wrapper raises (there is no CPU render path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rayzen_amd.renderer import RayZenError, Renderer
    with pytest.raises(RayZenError):
        Renderer(0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under rayzen_amd/ or include/ may reference it."""
    bad = []
    for base in ("rayzen_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"rz_oracle|from oracle|import oracle|oracle/|rzo\b", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_the_library_says_which_built_ins_it_was_compiled_with():
    """rz_math_flavour() (include/rayzen_hip.h; needs no GPU): the default build computes sin / cos / acos as the run of RayZen's own
    shader did (flavour 1: tests/test_glref.py), and the oracle's default is the same one."""
    from rayzen_amd import _lib
    from oracle import rzo
    import os
    f = _lib.hip().rz_math_flavour()
    assert f in (0, 1)
    if not os.environ.get("RAYZEN_HIP_SO"):
        assert f == 1
    import subprocess, sys
    out = subprocess.run([sys.executable, "-c", "from oracle import rzo; print(rzo.lib().rzo_get_math_flavour())"], capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env={k: v for k, v in os.environ.items() if k != "RZO_MATH_FLAVOUR"})
    assert out.stdout.strip() == "1", out.stderr
