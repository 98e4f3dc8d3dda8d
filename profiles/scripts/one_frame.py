"""Two frames of one BASELINE configuration with the default kernel (for rocprofv3 passes that should not pay for
bench.py's extras):  one_frame.py [c2|c4|c5|c5full|c2g|glassbunny|ref]   (c2 is the bench workload; c5 = C5 at 32 spp, c5full at its 128 spp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    W, H, SPP, B = 1920, 1080, 64, 4
    sc = S.bunny_scene(n=76, aspect=W / H)
elif which == "c4":
    W, H, SPP, B = 1920, 1080, 16, 4
    sc = S.instanced_scene(n=76, count=16, aspect=W / H)
elif which == "c5":
    W, H, SPP, B = 3840, 2160, 32, 8
    sc = S.stress_scene(n=289, aspect=W / H)
elif which == "c5full":            # BASELINE configs[4] as stated: 128 spp (the persistent, compacting launch; c5 at 32 spp is one workgroup per pixel pair)
    W, H, SPP, B = 3840, 2160, 128, 8
    sc = S.stress_scene(n=289, aspect=W / H)
elif which == "c2g":               # the C2 frame with a glass blob and a mirror cube added: the transparent-scene variant of the kernel
    W, H, SPP, B = 1920, 1080, 64, 4
    sc = S.bunny_scene(n=76, aspect=W / H, extras=True)
elif which == "glassbunny":        # the C2 frame with the bunny itself made of glass (material 3)
    W, H, SPP, B = 1920, 1080, 64, 4
    sc = S.bunny_scene(n=76, aspect=W / H, bunny_material=3)
elif which == "ref":               # RayZen's own workload (main.cpp:35-36, 356-384, 600; FS:675): 800x600, 1 spp, 5 bounces
    W, H, SPP, B = 800, 600, 1, 5
    sc = S.reference_scene(aspect=W / H)
else:
    raise SystemExit(which)
r = Renderer(0)
r.upload_scene(sc)
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP))
for _ in range(2):
    r.render()
r.sync()
print(which, [round(x, 2) for x in r.render_history_ms()])
r.close()
