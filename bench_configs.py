#!/usr/bin/env python3
"""Side harness: the BASELINE.json configurations other than the headline one, and their companions, on ONE GPU.

  c1       Cornell (16 tris) 256x256, 4 spp, 1 bounce
  c2close  the bench workload (configs[1]: same mesh, 1080p, 64 spp, 4 bounces) with the camera 0.9 units outside the mesh, so
           that most camera paths hit geometry (at the bench camera 64 % of them see only sky): the geometry-dominated companion
           of the headline number
  c2g      configs[1] + a glass blob and a mirror cube (the transparent-scene kernel); glassbunny: the mesh itself of glass
  c3       configs[2]'s frame (256 spp) whole on one GPU
  c4       16 instances of the bunny mesh (one shared BLAS), 1080p, 16 spp, 4 bounces, 30 frames; per frame the host
           re-derives the transforms, rebuilds the TLAS (librayzen_host) and rz_update()s instances + TLAS -- all inside
           the timed region, as RayZen's frame loop does (main.cpp:572)
  c4d      the same frames with rz_update_transforms: inverse, world AABBs and the TLAS rebuild run on the GPU
  c5       ~1M-triangle mesh, 3840x2160, 128 spp, 8 bounces (the per-GPU share of the 8-GPU config is 1/8 of the pixels;
           here one GPU renders the whole frame)
  c5d      the same with the BLAS built on the device (rz_build_blas) instead of by librayzen_host
  ref      RayZen's OWN workload (RayZen/src/main.cpp:35-36, 331-384, 600; fragment_shader.glsl:675): 800x600, 1 sample per pixel,
           bounce budget 1 on frame 0 and 5 afterwards, two lights, seven objects -- the cube floor, five ~1 k-triangle meshes (one
           mirror, one GLASS) and one EMPTY mesh (car.obj is absent from the reference) -- 100 frames, each with the reference's
           per-frame work in the timed region: updateDynamicBVHAndSSBOs (instances + TLAS rebuilt on the host, main.cpp:1123-1194)
           and the rz_update of instances / TLAS nodes / TLAS indices.  Reported beside the 60-Hz budget (16.7 ms).
bench.py stays the driver-facing benchmark (configs[1]); this prints one JSON line per config.  Every line carries the work
model of rayzen_amd/workmodel.py (the FLOOR of VALU lane slots the algorithm needs / what the chip offers in the kernel's
duration) and, when profiles/r05_<config>/pmc_rz_render_samples.json was captured from the loaded build, the measured fractions
beside it (issue fraction, VALU lane utilisation, useful lane fraction, TA busy, instruction-cache hit rate), so the table of
"which launch is furthest below its roof" comes from one command.
"""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))


ROOT = __import__("os").path.dirname(__import__("os").path.abspath(__file__))
PROFILE_ROUND = "r05"
PROFILE_OF = {"c4d": "c4", "c5": "c5", "c5d": "c5"}       # whose PMC file speaks for a row (same kernel launch)


def measured_fractions(name, kernel_s, wm):
    """The PMC-derived fractions of profiles/r04_<config>/, if that file belongs to the LOADED build."""
    import os
    from rayzen_amd import _lib as rzlib
    from rayzen_amd.workmodel import LANE_PEAK, executed_live_lane_valu
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_" + PROFILE_OF.get(name, name), "pmc_rz_render_samples.json")
    if not os.path.exists(path):
        return None
    pj = json.load(open(path))
    if pj.get("_source_hash") != rzlib.hip().rz_source_hash().decode():
        return {"counters_stale": os.path.relpath(path, ROOT)}
    insts = sum(pj[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"))
    live = executed_live_lane_valu(pj)
    out = {"issue_frac": round(insts / kernel_s / (1024 * 2.4e9 / 2.0), 4),
           "valu_lane_utilisation": round(pj["SQ_THREAD_CYCLES_VALU"] / (64.0 * pj["SQ_ACTIVE_INST_VALU"]), 3),
           "useful_lane_frac": round(live / kernel_s / LANE_PEAK, 4), "executed_live_lane_valu": int(live),
           "floor_over_executed": round(wm["lane_slots"] / max(live, 1.0), 4),
           "hbm_traffic_bytes": int(2 * pj["FETCH_SIZE"] * 1024 + pj["WRITE_SIZE"] * 1024) if "FETCH_SIZE" in pj and "WRITE_SIZE" in pj else None,
           "counters_from": os.path.relpath(path, ROOT)}
    if "TA_TA_BUSY_sum" in pj and "GRBM_GUI_ACTIVE" in pj:
        out["ta_busy"] = round(pj["TA_TA_BUSY_sum"] / (256.0 * pj["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if pj.get("SQC_ICACHE_REQ"):
        out["icache_hit_rate"] = round(pj.get("SQC_ICACHE_HITS", 0.0) / pj["SQC_ICACHE_REQ"], 5)
    out["floor_violated"] = bool(wm["lane_slots"] > live)       # the floor must lie below what the kernel executed on live lanes (enforced in tests/test_workmodel.py)
    return out


def main():
    which = sys.argv[1:] or ["c1", "ref", "ref16", "ref64", "c2close", "c2g", "glassbunny", "c3", "c4", "c4d", "c5", "c5d"]
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import Renderer, algorithmic_bytes, frame_params
    from rayzen_amd.workmodel import work_model
    for name in which:
        r = Renderer(0)
        t_build = time.perf_counter()
        if name in ("c4", "c4d"):
            sc, W, H, spp, b = S.named_config("c4")
            frames = 30
        elif name in ("c5", "c5d"):
            sc = S.stress_scene(n=289, aspect=3840 / 2160, blas_builder=r if name == "c5d" else None)
            W, H, spp, b, frames = 3840, 2160, 128, 8, 3
        elif name in S.NAMED_CONFIGS:
            sc, W, H, spp, b = S.named_config(name)
            frames = {"c1": 20, "ref": 100, "c3": 3}.get(name, 5)
        else:
            raise SystemExit(name)
        t_build = time.perf_counter() - t_build
        floor_xf = sc.arrays[S.BIND_INSTANCES]["transform"][0].copy() if name == "c4d" else None
        t_up = time.perf_counter()
        r.upload_scene(sc)
        r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), 1 if name == "ref" else b, spp))     # (RayZen's frame 0 has a bounce budget of 1: main.cpp:600)
        r.render(); r.sync()                                   # includes the one-time re-layout
        if name == "ref":
            r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
        t_up = time.perf_counter() - t_up
        cnt = r.render_counted()
        r.render_history_ms()
        t0 = time.perf_counter()
        for f in range(frames):
            if name == "c4":            # host TLAS rebuild + glBufferSubData-style update
                for oid, t in zip(sc.instance_ids, S.instanced_transforms(f + 1, 16)):
                    sc.set_transform(oid, t)
                sc.update_dynamic()
                r.update_dynamic(sc)
            elif name == "ref":         # the scene is static, but RayZen rebuilds instances + TLAS and re-uploads them every frame
                sc.update_dynamic()
                r.update_dynamic(sc)
            elif name == "c4d":         # device-side rebuild: only the transforms cross the bus
                import numpy as np
                r.update_transforms(np.stack([floor_xf] + S.instanced_transforms(f + 1, 16)))
            r.render()
        r.sync()
        dt = time.perf_counter() - t0
        kms = r.render_history_ms()
        kernel_s = min(kms) * 1e-3
        nl = max(1, len(sc.lights))
        wm = work_model(cnt, kernel_s)
        out = {"config": name, "scene": sc.name, "triangles": int(sc.arrays[S.BIND_TRIANGLES].shape[0]),
               "instances": int(sc.arrays[S.BIND_INSTANCES].shape[0]), "blas_depth": sc.max_blas_depth,
               "width": W, "height": H, "spp": spp, "bounces": b, "frames": frames,
               "ms_per_frame_wall": round(dt / frames * 1e3, 3), "kernel_ms": round(sum(kms) / len(kms), 3), "kernel_ms_min": round(min(kms), 3),
               "msamples_per_s": round(W * H * spp * frames / dt / 1e6, 1), "kernel": r.last_kernel_name(),
               "closest_hit_gqueries_per_s": round(cnt["traversals"] / kernel_s / 1e9, 2),
               "closest_hit_queries_per_path": round(cnt["traversals"] / max(cnt["samples"], 1), 3),
               "primary_hit_fraction": round(cnt["light_fetches"] / nl / max(cnt["samples"], 1), 4),
               "algorithmic_bytes_per_sample": round(algorithmic_bytes(cnt) / cnt["samples"], 1),
               "work_model": {k: wm[k] for k in ("lane_slots", "frac", "traversal_share")},
               "scene_build_s": round(t_build, 3), "upload_relayout_first_frame_s": round(t_up, 3)}
        mf = measured_fractions(name, kernel_s, wm)
        if mf:
            out["measured"] = mf
        if name == "ref":
            out["frames_per_s"] = round(frames / dt, 1)
            out["share_of_the_60_hz_budget"] = round(dt / frames / (1.0 / 60.0), 4)
        print(json.dumps(out), flush=True)
        r.close()


if __name__ == "__main__":
    main()
