"""Can RCCL take two ranks on ONE device?  (The boxes have one GPU; if it can, the group's ncclSend / ncclRecv gather runs between
two real ranks for the first time.)  Parent: never touches the GPU, starts two children, kills them after a time limit.
Child r: rz_group_create_rank(device 0, rank r of 2), render its tiles, gather on rank 0, compare with a single context."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(rank, uid_path):
    import numpy as np
    from rayzen_amd import dist as D
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import frame_params
    if rank == 0:
        uid = D.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        while not os.path.exists(uid_path):
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    print(f"[rank {rank}] joining", flush=True)
    g = D.Group.create_rank(0, rank, 2, uid)
    print(f"[rank {rank}] joined: transport {g.transport}", flush=True)
    sc = S.bunny_scene(n=8, extras=True)
    W, H, spp, b = 96, 54, 3, 4
    g.upload_scene(sc)
    g.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    for _ in range(2):
        g.render()
        g.reduce(0)
    g.sync()
    if rank == 0:
        from helpers import hip_render
        got = g.read_frame()
        ref = hip_render(sc, W, H, spp, b)
        print(f"[rank 0] gathered frame bit-identical to one context: {(got.view(np.uint32) == ref.view(np.uint32)).all()}; reduce ms {g.last_reduce_ms()}", flush=True)
    g.close()
    print(f"[rank {rank}] done", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), sys.argv[2])
        sys.exit(0)
    uid_path = f"/tmp/rz_uid_{os.getpid()}"
    env = dict(os.environ, NCCL_DEBUG="WARN")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), uid_path], env=env) for r in range(2)]
    t0 = time.time()
    while time.time() - t0 < 90 and any(p.poll() is None for p in ps):
        time.sleep(0.5)
    for p in ps:
        if p.poll() is None:
            print(f"[parent] killing pid {p.pid} (still running after 90 s)", flush=True)
            p.kill()
    print("[parent] exit codes", [p.wait() for p in ps], flush=True)
