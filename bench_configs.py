#!/usr/bin/env python3
"""Side harness: the BASELINE.json configurations other than the headline one, on ONE GPU.

  c1  Cornell (16 tris) 256x256, 4 spp, 1 bounce
  c4  16 instances of the bunny mesh (one shared BLAS), 1080p, 16 spp, 4 bounces, 30 frames; per frame the host
      re-derives the transforms, rebuilds the TLAS (librayzen_host) and rz_update()s instances + TLAS -- all inside
      the timed region, as RayZen's frame loop does (main.cpp:572)
  c4d the same frames with rz_update_transforms: inverse, world AABBs and the TLAS rebuild run on the GPU
  c5  ~1M-triangle mesh, 3840x2160, 128 spp, 8 bounces (the per-GPU share of the 8-GPU config is 1/8 of the pixels;
      here one GPU renders the whole frame)
  c5d the same with the BLAS built on the device (rz_build_blas) instead of by librayzen_host
  ref RayZen's OWN workload (RayZen/src/main.cpp:35-36, 331-384, 600; fragment_shader.glsl:675): 800x600, 1 sample per pixel, bounce
      budget 1 on frame 0 and 5 afterwards, two lights, seven objects -- the cube floor, five ~1 k-triangle meshes (one mirror,
      one GLASS) and one EMPTY mesh (car.obj is absent from the reference) -- 100 frames, each with the reference's per-frame
      work in the timed region: updateDynamicBVHAndSSBOs (instances + TLAS rebuilt on the host, main.cpp:1123-1194) and the
      rz_update of instances / TLAS nodes / TLAS indices.  Reported beside the 60-Hz budget (16.7 ms) the reference targets.
bench.py stays the driver-facing benchmark (configs[1]); this prints one JSON line per config.
"""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def main():
    which = sys.argv[1:] or ["c1", "ref", "c4", "c4d", "c5", "c5d"]
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import Renderer, algorithmic_bytes, frame_params
    for name in which:
        r = Renderer(0)
        t_build = time.perf_counter()
        if name == "c1":
            sc, W, H, spp, b, frames = S.cornell_scene(), 256, 256, 4, 1, 20
        elif name == "ref":
            sc, W, H, spp, b, frames = S.reference_scene(aspect=800 / 600), 800, 600, 1, 5, 100
        elif name in ("c4", "c4d"):
            sc, W, H, spp, b, frames = S.instanced_scene(n=76, count=16, aspect=1920 / 1080), 1920, 1080, 16, 4, 30
        elif name in ("c5", "c5d"):
            sc = S.stress_scene(n=289, aspect=3840 / 2160, blas_builder=r if name == "c5d" else None)
            W, H, spp, b, frames = 3840, 2160, 128, 8, 3
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
        out = {"config": name, "scene": sc.name, "triangles": int(sc.arrays[S.BIND_TRIANGLES].shape[0]),
               "instances": int(sc.arrays[S.BIND_INSTANCES].shape[0]), "blas_depth": sc.max_blas_depth,
               "width": W, "height": H, "spp": spp, "bounces": b, "frames": frames,
               "ms_per_frame_wall": round(dt / frames * 1e3, 3), "kernel_ms": round(sum(kms) / len(kms), 3),
               "msamples_per_s": round(W * H * spp * frames / dt / 1e6, 1), "kernel": r.last_kernel_name(),
               "algorithmic_bytes_per_sample": round(algorithmic_bytes(cnt) / cnt["samples"], 1),
               "scene_build_s": round(t_build, 3), "upload_relayout_first_frame_s": round(t_up, 3)}
        if name == "ref":
            out["frames_per_s"] = round(frames / dt, 1)
            out["share_of_the_60_hz_budget"] = round(dt / frames / (1.0 / 60.0), 4)
        print(json.dumps(out), flush=True)
        r.close()


if __name__ == "__main__":
    main()
