"""The hand-written scalar-unit descend loop (rz_trace.h: RZ_WALK_ASM; fragment_shader.glsl:426-452) names its registers:
s[80:95] the pair, v[6:17] the six packed plane products, v18-v25 temporaries.  The `asm` statement declares them clobbered, so
hipcc may neither keep a value in them across the block nor hand one of them to an operand -- this test CHECKS that on the built
library, and that every kernel that carries the loop carries it once per direction octant with the operand
permutation slab_finish<OCT> prescribes: for axis a, bit a of OCT set <=> the ray points down the axis <=> the NEAR plane is the
box's max plane, i.e. the product in the ODD register of the pair.  No GPU needed: the code object is disassembled here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FIXED_V = set(range(6, 26))
FIXED_S = set(range(80, 96))


def _regs(tok):
    """Register numbers an operand token names: v7 -> ('v', {7}); s[80:81] -> ('s', {80, 81}); anything else -> None."""
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    return None


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm install is not here")
    from rayzen_amd import build
    tmp = tmp_path_factory.mktemp("asmwalk")
    so = shutil.copy(build.HIP_SO, tmp / "librayzen_hip.so")
    subprocess.run([OBJDUMP, "--offloading", str(so)], cwd=tmp, check=True, capture_output=True)     # writes the bundles beside the copy
    cos = sorted((p for p in os.listdir(tmp) if "gfx950" in p), key=lambda p: -os.path.getsize(tmp / p))
    assert cos, "no gfx950 code object in the library"
    text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(tmp / cos[0])], check=True, capture_output=True, text=True).stdout
    return text.split("\n")


def _blocks(lines):
    """(kernel symbol, the instructions of one hand-written loop: from its fetch to the two s_mov that hand the child references out)."""
    out, sym = [], None
    i = 0
    while i < len(lines):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", lines[i])
        if m:
            sym = m.group(1)
        # (the signature of the hand-written loop: the fetch into s[80:95] with the first plane subtraction straight behind the wait --
        #  hipcc may give s[80:95] to a fetch of the C++ walk too)
        if "s_load_dwordx16 s[80:95]" in lines[i] and any("v_pk_add_f32 v[6:7], s[80:81]" in l for l in lines[i + 1:i + 4]):
            blk = []
            j = i
            while j < len(lines) and j < i + 80:
                ins = lines[j].split("//")[0].strip()
                if ins:
                    blk.append(ins)
                # the block ends with `s_mov_b32 <lenc>, s92` / `s_mov_b32 <renc>, s93` after the v_mov of the two entry distances
                if re.match(r"s_mov_b32 s\d+, s93$", ins) and len(blk) > 40:
                    break
                j += 1
            out.append((sym, blk))
            i = j
        i += 1
    return out


def test_the_hand_written_loop_keeps_its_registers_and_its_octant_permutations(disassembly):
    blocks = _blocks(disassembly)
    assert blocks, "the library carries no hand-written descend loop (RZ_ASM_WALK off?)"
    per_kernel = {}
    for sym, blk in blocks:
        # (1) every operand hipcc chose (the %[..] operands: ray origin and reciprocal, tLoc, the stack pointer and column, the masks,
        #     the cursor) lies OUTSIDE the registers the block names for itself
        for ins in blk:
            op, _, rest = ins.partition(" ")
            toks = [t.strip() for t in re.split(r",\s*", re.sub(r"\s+(neg_lo|neg_hi|op_sel|op_sel_hi):\[[^\]]*\]", "", rest)) if t.strip()]
            for t in toks:
                r = _regs(t)
                if r is None:
                    continue
                kind, nums = r
                fixed = FIXED_V if kind == "v" else FIXED_S
                inside = nums & fixed
                assert not inside or nums <= fixed, f"{sym}: `{ins}` straddles the block's fixed registers"
        named = {(k, n) for ins in blk for t in re.split(r"[ ,]+", ins) for (k, ns) in [(_regs(t) or (None, set()))] for n in ns}
        free_v = {n for k, n in named if k == "v"} - FIXED_V
        free_s = {n for k, n in named if k == "s"} - FIXED_S
        assert free_v and free_s and not (free_v & FIXED_V) and not (free_s & FIXED_S)
        # (2) the octant: which plane product is NEAR (max3 of the near planes = tmin), per axis, for both boxes
        mx = [re.match(r"v_max3_f32 v(\d+), v(\d+), v(\d+), v(\d+)", i) for i in blk]
        mn = [re.match(r"v_min3_f32 v(\d+), v(\d+), v(\d+), v(\d+)", i) for i in blk]
        mx = {int(m.group(1)): tuple(int(x) for x in m.groups()[1:]) for m in mx if m}
        mn = {int(m.group(1)): tuple(int(x) for x in m.groups()[1:]) for m in mn if m}
        assert set(mx) == {19, 22} and set(mn) == {20, 23}, (sym, mx, mn)
        octs = []
        for near, far, base in ((mx[19], mn[20], 6), (mx[22], mn[23], 12)):
            o = 0
            for a in range(3):
                lo, hi = base + 2 * a, base + 2 * a + 1         # (min plane product, max plane product) of axis a
                assert {near[a], far[a]} == {lo, hi}, (sym, near, far)
                if near[a] == hi:
                    o |= 1 << a
            octs.append(o)
        assert octs[0] == octs[1], f"{sym}: the two boxes of a pair are tested in different octant forms {octs}"
        per_kernel.setdefault(sym, []).append(octs[0])
    for sym, octs in per_kernel.items():
        # (a kernel may hold several inlined copies of the walk -- the speculating group code holds two: every copy brings all eight forms)
        n = len(octs) // 8
        assert n >= 1 and sorted(octs) == sorted(list(range(8)) * n), f"{sym}: octant forms {sorted(octs)} (expected each of 0..7 equally often)"
    assert len(per_kernel) >= 8         # the non-counting, non-overflow instantiations of rz_render_samples (+ the pixel kernel)
