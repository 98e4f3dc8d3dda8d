import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer
sc = S.instanced_scene(n=76, count=16, aspect=1920/1080)
r = Renderer(0); r.upload_scene(sc); r.render_scene(sc, 1920, 1080, 4, 4)
for kw in (dict(fps=60.0), dict(fps=60.0, show_lights=True, show_bvh=True, bvh_mode=0), dict(fps=60.0, show_bvh=True, bvh_mode=1, selected_blas=3, selected_tri=100)):
    for _ in range(5): r.present(**kw)
r.close()
