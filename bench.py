#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing render loop on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one frame of synthetic input: BASELINE.json configs[1], the
~69k-triangle "bunny" scene (procedural stand-in, see rayzen_amd/scene.py) at 1920x1080, 4 bounces, 64 samples per
pixel per GPU.  At N > 1 the frame's 8x8-pixel tiles are dealt round-robin to the ranks and ONE exchange step per step
lands the frame on rank 0: one RCCL reduce(SUM) of the accumulation buffers (`--transport reduce`, the default: what
BASELINE.json's north_star names) or a gather of the ranks' own tiles (`--transport gather`) -- both through the C-ABI's
multi-GPU group (include/rayzen_hip.h: rz_group_*), enqueued on each member's render stream.  Two ways to start it:
  * `python bench.py --gpus N` as ONE plain process (no launcher, WORLD_SIZE unset): rz_group_create(N) -- N contexts on
    the node's first N devices and their communicators from ncclCommInitAll; nothing is re-executed, no torch.distributed;
  * under torch.distributed.run (WORLD_SIZE = N): one process per GPU, rz_group_create_rank with an id broadcast over
    torch.distributed (ncclCommInitRank).
The JSON says which ran (`config.parallelism`), how many ranks RCCL saw (`rccl_ranks`) and what the reduce cost
(`reduce_ms`: HIP events around it on the root's stream).  Default N > 1 workload: every rank renders 64*N spp of its own
pixels (per-GPU work constant: "weak"); `--spp-total T` fixes the frame at T spp instead ("strong"; `--spp-total 256` at
N = 8 is BASELINE configs[2], the C3 line).  Every N > 1 line ALSO carries `configs2`: three extra steps of the 256-spp frame
of BASELINE configs[2] on the same N GPUs (strong scaling of that frame; at N = 8 it IS configs[2]).
A run cannot die on the exchange step: a group whose RCCL communicator cannot be made, or whose first exchange fails in the
warm-up, is rebuilt in the same process (never re-executed) on the next transport down -- rccl gather -> rccl reduce -> plain
device copies between the members' GPUs (one-process mode) / torch.distributed.reduce (launcher mode) -- and the line says
which ran and why (`transport`, `transport_fallback`).
Scene buffers are resident in HBM before the timed region; nothing is skipped inside it.

Rank 0 prints ONE JSON line.  `value` = total camera paths (pixels x spp) / wall time of the K timed steps (max over
ranks).  `roofline` prices the render kernel against what bounds it -- the SIMDs' instruction issue rate (measured:
profiles/r02_valu_issue, profiles/r02_issue_sensitivity) -- with the HBM view (algorithmic bytes, counter traffic)
kept beside it.  `cpu_baseline` = the oracle (a CPU port of the same path) timed on a bounded sample of the same
frame, which doubles as a full-size parity check of those pixels.
"""
import argparse
import json
import os
import sys
import time

# RCCL / cross-process device memory on this pool need dmabuf IPC (the host driver supports nothing else); exported on
# the boxes already -- set here too, before any HIP runtime is loaded, for whoever starts the bench from a bare environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
SIMDS = 256 * 4                 # 256 CUs x 4 SIMD-32
MAX_CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md "Max clock"
ISSUE_CYCLES_PER_INST = 2.0     # a SIMD issues at most one wave64 instruction per 2 cycles (profiles/r02_valu_issue)
ISSUE_PEAK_GINST = SIMDS * MAX_CLOCK_HZ / ISSUE_CYCLES_PER_INST / 1e9
PMC_JSON = os.path.join(ROOT, "profiles", "r05_c2_kernel", "pmc_rz_render_samples.json")
# Measured issue cost (SIMD cycles per wave64 instruction with four waves resident, profiles/r02_valu_issue/valu_issue.txt)
# of the VALU classes the SQ counters tell apart; "other" = comparisons, selects, min / max, moves, lane reads.  The f32
# add / mul / fma class mixes 2-cycle scalar-free forms with 3.5-cycle packed and SGPR-operand forms (2.6 assumed), int32
# mixes 1.9-cycle adds / ands with 3.4-cycle shifts (2.7 assumed); non-vector instructions are priced at the 2.0 cycles
# they cost when interleaved with vector ones.  A MODEL of how busy this instruction mix keeps the SIMDs, not a counter.
MIX_COST = {"SQ_INSTS_VALU_TRANS_F32": 6.57, "SQ_INSTS_VALU_TRANS_F64": 12.7, "SQ_INSTS_VALU_ADD_F64": 3.52,
            "SQ_INSTS_VALU_MUL_F64": 3.52, "SQ_INSTS_VALU_FMA_F64": 3.52, "SQ_INSTS_VALU_CVT": 3.5, "SQ_INSTS_VALU_INT64": 3.45,
            "SQ_INSTS_VALU_INT32": 2.7, "SQ_INSTS_VALU_ADD_F32": 2.6, "SQ_INSTS_VALU_MUL_F32": 2.6, "SQ_INSTS_VALU_FMA_F32": 2.6}
MIX_COST_OTHER_VALU, MIX_COST_NON_VALU = 3.4, 2.0


def mix_model(pj, kernel_s, clock_hz):
    """Issue cycles this launch's instruction mix needs at the measured per-class costs / SIMD cycles it had."""
    if not all(k in pj for k in MIX_COST):
        return None
    classed = sum(pj[k] for k in MIX_COST)
    valu_cycles = sum(pj[k] * c for k, c in MIX_COST.items()) + max(0.0, pj["SQ_INSTS_VALU"] - classed) * MIX_COST_OTHER_VALU
    other = sum(pj[k] for k in INST_COUNTERS if k != "SQ_INSTS_VALU")
    have = kernel_s * clock_hz * SIMDS
    return {"valu_cycles": int(valu_cycles), "non_valu_cycles": int(other * MIX_COST_NON_VALU), "simd_cycles_available": int(have),
            "busy_frac_valu_only": round(valu_cycles / have, 3), "busy_frac": round((valu_cycles + other * MIX_COST_NON_VALU) / have, 3),
            "clock_ghz_assumed": round(clock_hz / 1e9, 2),
            "note": "mix-weighted issue model (per-class costs from profiles/r02_valu_issue): the instruction mix of this launch keeps "
                    "the SIMDs busy for this fraction of the kernel's duration; a model, not a counter"}


# ---- work model (rayzen_amd/workmodel.py) ----------------------------------------------------------
# The issue fraction above rises when a kernel executes MORE instructions; it cannot compare designs.  The work model prices
# what the reference ALGORITHM needs for the frame that was rendered -- the units rz_render_counted tallies for THIS launch -- as
# a FLOOR of VALU issue slots per lane (every instruction one slot, best known bit-exact forms), so that it can never exceed
# what the kernel executed on live lanes; it rises only when the frame gets faster.
from rayzen_amd.workmodel import LANE_PEAK, executed_live_lane_valu, work_model      # noqa: E402


INST_COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                 "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # workload overrides (the defaults ARE the BASELINE config; anything else is for development)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel per GPU (weak scaling)")
    ap.add_argument("--spp-total", type=int, default=0,
                    help="fix the frame's samples per pixel whatever N is (strong scaling; 256 at N = 8 is configs[2])")
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--mesh-n", type=int, default=76, help="bunny stand-in has 12*n*n triangles")
    ap.add_argument("--obj", default=None, help="render this OBJ instead of the procedural stand-in (default: assets/bunny.obj if present)")
    ap.add_argument("--backend", choices=["auto", "pixel"], default="auto",
                    help="render pipeline: auto = the library's default")
    ap.add_argument("--reduce", choices=["group", "torch", "torch-gloo"], default="group",
                    help="group = the C-ABI's rz_group (RCCL over xGMI, the product path); torch = torch.distributed.reduce on the "
                         "ranks' device buffers (backend nccl = RCCL; also the automatic fallback if the group cannot be formed); "
                         "torch-gloo = rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share devices, "
                         "reduce staged through the host)")
    ap.add_argument("--transport", choices=["reduce", "gather"], default="reduce",
                    help="the group's exchange step: reduce = ONE ncclReduce(SUM) of the accumulation buffers (north_star; default), "
                         "gather = each rank's own tiles sent straight to the root (ncclSend / ncclRecv; 1 / N of the bytes)")
    ap.add_argument("--no-configs2", action="store_true", help="skip the extra BASELINE configs[2] (256 spp) leg of an N > 1 run")
    ap.add_argument("--loopback", action="store_true",
                    help="REHEARSAL of `--gpus N` as one plain process on a box with fewer GPUs: the N members of the group share "
                         "device 0 and device copies stand in for the links (RZ_GROUP_LOOPBACK); everything else -- the dealing of "
                         "tiles, packing, the root's scatter, the timing -- is the code an N-GPU run executes.  Not a scaling number.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (default: min(16, usable cores) = the box's CPU share)")
    ap.add_argument("--cpu-bands", type=int, default=36, help="oracle sample: this many 8-row bands of the frame")
    ap.add_argument("--cpu-full-frame", action="store_true", help="also time the oracle on the WHOLE frame (about 6 s on 16 threads)")
    return ap.parse_args(argv)


def launch_mode(gpus, environ):
    """How `--gpus N` is served: 'single' (N = 1), 'ranks' (a launcher started one process per GPU: WORLD_SIZE = N), or
    'local-group' (ONE plain process drives N devices through rz_group_create -- what `python3 bench.py --gpus N` gets
    when nothing launched it; no process is ever re-executed)."""
    world = int(environ.get("WORLD_SIZE", "1"))
    if world > 1:
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}")
        return "ranks"
    return "local-group" if gpus > 1 else "single"


def main(argv=None):
    a = parse(argv)
    import numpy as np

    mode = launch_mode(a.gpus, os.environ)
    world = a.gpus
    rank = int(os.environ.get("RANK", "0")) if mode == "ranks" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if mode == "ranks" else 0

    from rayzen_amd import _lib as rzlib
    from rayzen_amd import build as rzbuild
    from rayzen_amd import dist as rzdist
    from rayzen_amd import scene as S
    from rayzen_amd.renderer import Renderer, algorithmic_bytes, frame_params

    if mode == "local-group" and a.reduce != "group":
        raise SystemExit("--reduce torch / torch-gloo need one process per GPU: start bench.py with torch.distributed.run")
    if a.loopback and mode != "local-group":
        raise SystemExit("--loopback rehearses `--gpus N` (N > 1) started as one plain process")

    torch = dist = dev = None
    use_group = world > 1 and a.reduce == "group"
    use_nccl = mode == "ranks" and a.reduce in ("group", "torch")
    dev_index = 0
    if mode == "ranks":
        # torch FIRST, and on its device, before librayzen_hip.so is loaded: the torch wheel bundles its own HIP runtime
        # (torch/lib/libamdhip64.so), and a process that has /opt/rocm's mapped already -- which loading the library first would
        # do -- leaves torch with "No HIP GPUs are available" (profiles/scripts/hip_runtime_order.py; found in round 4: the
        # launcher mode had not run on a GPU box since the device count below moved in front of this block).  Loaded second,
        # the library binds to the runtime torch brought -- one runtime in the process, and one RCCL (rz_group.hip).
        import torch
        import torch.distributed as dist
        if torch.cuda.device_count() <= 0:
            raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
        dev_index = local_rank if use_nccl else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        torch.zeros(1, device=dev)      # (the runtime is up before anything else asks for it)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_nccl:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # group mode: barrier + max-over-ranks only
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    ndev = rzlib.hip().rz_device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if mode == "local-group" and ndev < world and not a.loopback:
        raise SystemExit(f"--gpus {world} but only {ndev} HIP device(s) are visible to this process")

    W, H, bounces = a.width, a.height, a.bounces
    spp_total = a.spp_total if a.spp_total > 0 else a.spp * world   # every rank renders ALL samples of its own pixels
    scaling = "strong" if a.spp_total > 0 else "weak"
    # a real bunny.obj dropped into assets/ replaces the procedural stand-in; none ships with the repo
    obj = a.obj or (os.path.join(ROOT, "assets", "bunny.obj") if os.path.exists(os.path.join(ROOT, "assets", "bunny.obj")) else None)
    sc = S.bunny_scene(n=a.mesh_n, aspect=W / H, obj_path=obj)
    flags = {"auto": 0, "pixel": 1}[a.backend]
    fp = frame_params(sc.camera, W, H, len(sc.lights), bounces, spp_total, 0, rank, world)

    group = None
    transport_fallback = []         # why the run is not on the transport it was asked for (empty: it is)

    def local_group(kind):
        """One process, N devices.  kind: 'rccl' (contexts + communicators from ncclCommInitAll; the exchange step asked for),
        'copies' (no communicator: tile gather by device copies between the members' GPUs), 'loopback' (rehearsal on device 0)."""
        if kind == "loopback":
            return rzdist.Group.create(world, [0] * world, flags | rzdist.GROUP_LOOPBACK)
        if kind == "copies":
            return rzdist.Group.create(world, list(range(world)), flags | rzdist.GROUP_LOOPBACK)
        g = rzdist.Group.create(world, None, flags)
        try:
            g.set_transport(a.transport)
        except Exception as e:      # (a bound RCCL without ncclSend / ncclRecv: the reduce is what it has)
            transport_fallback.append(f"{a.transport} -> reduce: {e}")
            g.set_transport("reduce")
        return g

    if mode == "local-group":
        if a.loopback:
            group = local_group("loopback")
        else:
            try:
                group = local_group("rccl")
            except Exception as e:
                print(f"[bench] the RCCL group cannot be made ({e}); the exchange step falls back to device copies between the GPUs", file=sys.stderr)
                transport_fallback.append(f"rccl -> device copies: {e}")
                group = local_group("copies")
    elif use_group:
        # Two steps, so that a rank which cannot even bind RCCL or make its context never leaves the others waiting inside
        # ncclCommInitRank: (1) every rank probes, all agree; (2) rank 0 makes the communicator id, everybody receives it over
        # the launcher's own process group and joins.
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            rzdist.rccl_version()
            Renderer(dev_index, flags).close()
        except Exception as e:
            print(f"[bench] rank {rank}: cannot bind RCCL / make a context ({e}); falling back to torch.distributed.reduce", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        uid = torch.zeros(129, dtype=torch.uint8, device=dev)       # 128 id bytes + 1 "rank 0 could make it" flag
        if int(ok.item()) == 1 and rank == 0:
            try:
                uid[:128].copy_(torch.frombuffer(bytearray(rzdist.unique_id()), dtype=torch.uint8))
                uid[128] = 1
            except Exception as e:
                print(f"[bench] rz_group_unique_id failed ({e}); falling back to torch.distributed.reduce", file=sys.stderr)
        dist.broadcast(uid, 0)
        if int(ok.item()) == 1 and int(uid[128].item()) == 1:
            # (3) join -- and agree on the OUTCOME too: a rank whose rz_group_create_rank fails after the
            # communicator's rendezvous (or on every rank alike) must not leave the others rendering into a group it is not in
            made = torch.ones(1, dtype=torch.int32, device=dev)
            try:
                group = rzdist.Group.create_rank(dev_index, rank, world, bytes(uid[:128].cpu().numpy().tobytes()), flags)
            except Exception as e:
                print(f"[bench] rank {rank}: rz_group_create_rank failed ({e}); falling back to torch.distributed.reduce", file=sys.stderr)
                group = None
                made.zero_()
            dist.all_reduce(made, op=dist.ReduceOp.MIN)
            if int(made.item()) != 1:
                if group is not None:
                    group.close()
                    group = None
                use_group = False
                transport_fallback.append("rz_group_create_rank failed on some rank -> torch.distributed.reduce")
            else:
                # every rank must use the SAME exchange step: the wish is the command line's, a rank whose RCCL cannot
                # do it says so, and all ranks take the minimum (0 = reduce, 1 = gather)
                can = torch.ones(1, dtype=torch.int32, device=dev)
                if a.transport == "gather":
                    try:
                        group.set_transport("gather")
                    except Exception as e:
                        print(f"[bench] rank {rank}: no tile gather here ({e})", file=sys.stderr)
                        can.zero_()
                else:
                    can.zero_()
                dist.all_reduce(can, op=dist.ReduceOp.MIN)
                group.set_transport("gather" if int(can.item()) == 1 else "reduce")
                if a.transport == "gather" and int(can.item()) != 1:
                    transport_fallback.append("gather -> reduce: some rank's RCCL has no ncclSend / ncclRecv")
        else:
            use_group = False
            transport_fallback.append("RCCL could not be bound / no context / no communicator id on some rank -> torch.distributed.reduce")
    st = {"group": group, "r": None, "members": None, "accum": None, "frame": None}

    def adopt():
        """(Re)build the per-process render state around st['group'] (None: a plain context, with torch.distributed.reduce
        as the exchange step in launcher mode)."""
        g = st["group"]
        if g is not None:
            g.upload_scene(sc)
            g.set_frame(fp)                 # tile_rank / tile_nranks are filled in by the group
            st["members"] = [g.member(i) for i in range(g.local_count)]
            st["r"] = st["members"][0]
            st["accum"] = st["frame"] = None
            return
        rr = Renderer(dev_index, flags)
        st["r"], st["members"] = rr, [rr]
        rr.upload_scene(sc)
        if mode == "ranks":
            acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)    # zero outside this rank's tiles
            stream = torch.cuda.Stream(dev)     # the kernels, the reduce and the fences all order on this stream
            torch.cuda.set_stream(stream)
            rr.set_stream(stream.cuda_stream)
            rr.bind_accum(acc.data_ptr(), acc.numel() * 4)
            # `accum` is this rank's private buffer; each step copies it to `frame` and reduces THAT
            st["accum"], st["frame"] = acc, torch.empty_like(acc)
        rr.set_frame(fp)

    def step():
        g = st["group"]
        if g is not None:
            g.render()                      # asynchronous on the members' streams
            g.reduce(0)                     # the frame's exchange step, enqueued behind the kernels
            return
        st["r"].render()
        if mode == "ranks":
            st["frame"].copy_(st["accum"])
            if use_nccl:                    # torch path: one RCCL reduce(SUM) of the 33 MB frame
                rzdist.reduce_accum(st["frame"], dst=0)
            else:                           # rehearsal: same data flow, staged through the host
                host = st["frame"].cpu()
                rzdist.reduce_accum(host, dst=0)
                if rank == 0:
                    st["frame"].copy_(host)

    def fence():
        # drain this process's own work first, so that the group's RCCL communicator and torch's (the barrier) are never
        # in flight together; then the barrier, then the device once more (the barrier itself runs on the GPU)
        if st["group"] is not None:
            st["group"].sync()
        else:
            st["r"].sync()
        if mode == "ranks":
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)

    adopt()
    # PROBE (untimed, before the warm-up): one whole step through the exchange.  A group whose exchange fails here is replaced,
    # in this process, by the next transport down; nothing is re-executed, and the line says what happened.  (rz_group_reduce
    # itself already falls back from the gather to the reduce; what arrives here is a reduce that failed too.)
    if st["group"] is not None and not a.loopback:
        ok_probe, why = True, ""
        try:
            step()
            st["group"].sync()
        except Exception as e:
            ok_probe, why = False, str(e)
        if mode == "ranks":
            okt = torch.tensor([1 if ok_probe else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) != 1:
                print(f"[bench] rank {rank}: the group's exchange step failed on some rank ({why or 'not here'}); falling back to torch.distributed.reduce", file=sys.stderr)
                transport_fallback.append(f"rz_group exchange failed in the probe step ({why or 'on another rank'}) -> torch.distributed.reduce")
                try:
                    st["group"].close()
                except Exception:
                    pass
                st["group"], use_group = None, False
                adopt()
        elif not ok_probe:
            print(f"[bench] the group's exchange step failed ({why}); rebuilding the group on device copies between the GPUs", file=sys.stderr)
            transport_fallback.append(f"rccl exchange failed in the probe step ({why}) -> device copies")
            try:
                st["group"].close()
            except Exception:
                pass
            st["group"] = local_group("copies")
            adopt()
        tp = st["group"].transport if st["group"] is not None else ""
        if "fallback" in tp:
            transport_fallback.append(tp)
    group, r, members = st["group"], st["r"], st["members"]

    # untimed instrumented launches: exact algorithmic unit counts of this process's launches (all members of a local group)
    counters = None
    for m in members:
        cm = m.render_counted()
        counters = cm if counters is None else {k: counters[k] + cm[k] for k in cm}
    alg_bytes = algorithmic_bytes(counters)
    fence()

    for _ in range(a.warmup):
        step()
    fence()
    for m in members:
        m.render_history_ms()           # drain: only the timed launches remain in the event rings
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # GPU duration of each timed launch: HIP event pairs recorded on the launch stream inside the timed region (of a local
    # group: the member whose launches took longest)
    member_ms = [float(np.mean(m.render_history_ms())) for m in members]
    kms = max(member_ms)
    reduce_ms = group.last_reduce_ms() if group is not None else None

    rank_ms = None
    if mode == "ranks":
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if use_nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own kernel time and its share of the reduce (HIP events on its render stream), gathered so that ONE line
        # splits a step into kernel and collective: [kernel ms, reduce ms or -1] per rank
        mine = torch.tensor([kms, (reduce_ms[1] if reduce_ms is not None else -1.0)], dtype=torch.float64, device=dev if use_nccl else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_ms = [[float(x[0]), float(x[1])] for x in allr]

    # the frame of the LAST timed step, before anything else is rendered
    final = None
    if rank == 0:
        if group is not None:
            final = group.read_frame()
        elif mode == "ranks":
            final = st["frame"].cpu().numpy()
        else:
            final = r.read_accum()

    # N > 1: BASELINE configs[2] beside the headline -- the 256-spp frame on these N GPUs (strong scaling of THAT frame; at
    # N = 8 it is configs[2] literally).  Three steps behind one warm-up step, same fences, same max over ranks.
    configs2 = None
    if world > 1 and not a.no_configs2 and not (a.spp_total == 256):
        fp2 = frame_params(sc.camera, W, H, len(sc.lights), bounces, 256, 0, rank, world)
        (group.set_frame if group is not None else r.set_frame)(fp2)
        step()
        fence()
        for m in members:
            m.render_history_ms()
        c0 = time.perf_counter()
        for _ in range(3):
            step()
        fence()
        e2 = time.perf_counter() - c0
        k2 = max(float(np.mean(m.render_history_ms())) for m in members)
        if mode == "ranks":
            t = torch.tensor([e2, k2], dtype=torch.float64, device=dev if use_nccl else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2, k2 = float(t[0].item()), float(t[1].item())
        configs2 = {"workload": f"BASELINE configs[2]'s frame (same scene, {W}x{H}, 256 spp) tile-sharded across {world} GPU(s)"
                                + ("" if world == 8 else f" -- configs[2] itself names 8 GPUs; this is its frame on {world}"),
                    "is_baseline_configs2": bool(world == 8 and (W, H, bounces) == (1920, 1080, 4)),
                    "scaling": "strong", "spp_total": 256, "steps": 3, "warmup": 1,
                    "ms_per_step": round(e2 / 3 * 1e3, 3), "value": round(W * H * 256 * 3 / e2 / 1e6, 3), "unit": "Msamples/s",
                    "slowest_kernel_ms": round(k2, 3)}
        if rank == 0:
            f2 = group.read_frame() if group is not None else (st["frame"].cpu().numpy() if mode == "ranks" else r.read_accum())
            configs2["frame_check"] = {"every_pixel_has_256_samples": bool((f2[..., 3] == 256.0).all()),
                                       "finite_and_nonnegative": bool(np.isfinite(f2).all() and (f2 >= 0).all())}

    total_samples = W * H * spp_total * a.steps
    value = total_samples / elapsed / 1e6
    ntri = int(sc.arrays[S.BIND_TRIANGLES].shape[0])
    cam = sc.camera

    # what BASELINE.json calls this frame: configs[1] is the 64-spp frame on ONE GPU, configs[2] the 256-spp frame on 8; a
    # weak-scaling frame of 64 x N spp is neither, and is labelled as what it is
    std = (W, H, bounces, a.mesh_n) == (1920, 1080, 4, 76)
    if std and world == 1 and spp_total == 64:
        which_config = "configs[1]"
    elif std and world == 8 and spp_total == 256:
        which_config = "configs[2]"
    elif std and scaling == "weak" and a.spp == 64:
        which_config = (f"configs[1]'s scene and per-GPU work, WEAK scaling ({spp_total} spp per frame = 64 spp-equivalents per GPU; "
                        f"BASELINE lists no {spp_total}-spp frame: configs[2], the 256-spp frame, is timed beside it as `configs2`)")
    else:
        which_config = "configs[1]'s scene (development override)"
    out = {
        "metric": "Msamples/s (rays x spp / s) at 1080p", "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{which_config}: {'OBJ mesh ' + os.path.basename(obj) if obj else 'bunny stand-in'} ({sc.name}, {ntri} tris incl. "
                               f"floor) {W}x{H}, {spp_total} spp per frame ({spp_total // world if scaling == 'strong' else a.spp} "
                               f"spp-equivalents of work per GPU), {bounces} bounces, 2 lights",
                   "width": W, "height": H, "spp_per_gpu": a.spp, "spp_total": spp_total, "bounces": bounces, "triangles": ntri,
                   # SURVEY 8(d) put the camera at (0,0,3) in front of an 8-unit mesh, i.e. INSIDE it; this is the scene used instead
                   "camera": {"position": [float(x) for x in cam.position], "target_dir": [float(x) for x in cam.target],
                              "fov_deg": cam.fov}, "mesh_radius": 2.8, "mesh_centre": [0.0, 2.0, 0.0],
                   "floor": "cube scaled (8, 0.5, 8) at y = -3 (main.cpp:378)",
                   "parallelism": f"tiles8x8-roundrobin-x{world}" + ((("+rz_group-" + group.transport + "(" + (("one process, no communicator" if a.loopback else "one process, ncclCommInitAll") if mode == "local-group" else "one process per GPU, ncclCommInitRank") + ")")
                                                                     if use_group else ("+torch-rccl-reduce" if use_nccl else "+gloo-reduce(rehearsal)")) if world > 1 else "")},
    }
    if world > 1:
        # what RCCL saw: the size of the library's communicator (0: the reduce went through torch.distributed instead), and
        # the GPU time of the LAST timed step's reduce by HIP events on the root's stream -- it starts when the root's own
        # kernel ends, so it holds the wait for the slowest rank as well as the transfer (tile gather: 1 / N of the 33-MB
        # frame per member, straight to the root; RZ_GROUP_TRANSPORT=reduce: one ncclReduce of the whole buffers)
        out["rccl_ranks"] = group.size if group is not None else 0
        out["transport"] = group.transport if group is not None else "torch.distributed.reduce"
        out["transport_requested"] = a.transport
        if transport_fallback:
            out["transport_fallback"] = transport_fallback
        if configs2 is not None:
            out["configs2"] = configs2
        out["launch"] = mode + ("(loopback REHEARSAL: all members on device 0 -- not a scaling number)" if a.loopback else "")
        if reduce_ms is not None:
            out["reduce_ms"] = round(reduce_ms[0] if reduce_ms[0] >= 0 else reduce_ms[1], 3)
            out["reduce_ms_note"] = "HIP events around rz_group_reduce on the root member's stream (last timed step)"
        out["kernel_ms_per_member"] = [round(x, 3) for x in member_ms]
        if rank_ms is not None:
            out["kernel_ms_per_rank"] = [round(k, 3) for k, _ in rank_ms]
            out["reduce_ms_per_rank"] = [round(r_, 3) for _, r_ in rank_ms]
        # how one step divides: the slowest rank's kernel, then the reduce behind it; what is left is launch and fence overhead
        slow_k = max([k for k, _ in rank_ms]) if rank_ms is not None else kms
        red = out.get("reduce_ms")
        out["step_split_ms"] = {"slowest_kernel": round(slow_k, 3), "reduce": red, "step": out["ms_per_step"],
                                "other": round(out["ms_per_step"] - slow_k - (red or 0.0), 3)}
    if rank == 0:
        # size-independent check of the sharding + reduce: every pixel of the final frame received exactly
        # spp_total samples (a pixel rendered twice or not at all by the tile deal would show here)
        out["frame_check"] = {"every_pixel_has_spp_total_samples": bool((final[..., 3] == float(spp_total)).all()),
                              "finite_and_nonnegative": bool(np.isfinite(final).all() and (final >= 0).all())}
        nl = max(1, len(sc.lights))
        kernel_s = kms * 1e-3
        out["work"] = {"camera_paths_per_launch": counters["samples"],
                       "closest_hit_queries_per_path": round(counters["traversals"] / max(counters["samples"], 1), 3),
                       "closest_hit_gqueries_per_s": round(counters["traversals"] / kernel_s / 1e9, 2),
                       # lighting runs once per primary hit and fetches every light: light_fetches = lights x primary hits
                       "primary_hit_fraction": round(counters["light_fetches"] / nl / max(counters["samples"], 1), 4),
                       "kernel_msamples_per_s": round(counters["samples"] / kernel_s / 1e6, 2)}
        alg_gbps = alg_bytes / kernel_s / 1e9
        hbm = {"algorithmic_bytes_per_launch": int(alg_bytes),
               "algorithmic_bytes_per_sample": round(alg_bytes / max(counters["samples"], 1), 1),
               "algorithmic_GBps": round(alg_gbps, 1), "algorithmic_over_peak": round(alg_gbps / HBM_PEAK_GBPS, 3),
               "note": "algorithmic bytes = what RayZen's shader would read from its SSBOs for this frame (SURVEY 8d), counted "
                       "exactly by the instrumented launch; the 6 MB scene is cache resident, so this exceeds the HBM peak and "
                       "bounds nothing -- the counter traffic below is what HBM actually moved"}
        # (one process driving N devices: `counters` is the sum over the members, so the lane slots on offer are N chips')
        wm = work_model(counters, kernel_s, n_chips=len(members))
        roof = {"bound": "inst-issue", "achieved": None, "peak": round(ISSUE_PEAK_GINST, 1), "unit": "Gwave-inst/s", "frac": None,
                "traffic": None, "kernel": r.last_kernel_name(), "kernel_ms": round(kms, 3), "hbm": hbm,
                "peak_note": f"{SIMDS} SIMDs x {MAX_CLOCK_HZ / 1e9} GHz / {ISSUE_CYCLES_PER_INST} cycles per wave64 instruction "
                             "(measured issue limit of a gfx950 SIMD, vector and scalar instructions alike: profiles/r02_valu_issue)"}
        # instruction counts and HBM traffic come from rocprofv3 --pmc passes (they cannot be read in-process); they are
        # deterministic per (build, workload), so they are only used when they were captured from THIS build and workload
        # ... which is decided on the hash compiled into the LOADED library (rz_source_hash), not on the files beside it
        from rayzen_amd import _lib as rzlib
        src_hash = rzlib.hip().rz_source_hash().decode()
        tree_hash = rzbuild.source_hash()
        out["library"] = {"path": os.path.relpath(rzlib.HIP_SO, ROOT), "source_hash": src_hash[:16], "tree_source_hash": tree_hash[:16],
                          "built_from_this_tree": src_hash == tree_hash}
        if world == 1 and os.path.exists(PMC_JSON):
            try:
                pj = json.load(open(PMC_JSON))
                same = pj.get("_source_hash") == src_hash and pj.get("_workload") == [W, H, a.spp, bounces, a.mesh_n] and not obj
                if same:
                    insts = sum(pj[k] for k in INST_COUNTERS)
                    ach = insts / kernel_s / 1e9
                    traffic = int(2 * pj["FETCH_SIZE"] * 1024 + pj["WRITE_SIZE"] * 1024)   # gfx950: FETCH_SIZE under-reports wide reads 2x (MI355X_MICROARCH.md)
                    roof.update({"achieved": round(ach, 1), "frac": round(ach / ISSUE_PEAK_GINST, 4), "traffic": traffic,
                                 "wave_instructions_per_launch": int(insts),
                                 "instruction_mix": {k[9:].lower(): int(pj[k]) for k in INST_COUNTERS},
                                 "valu_lane_utilisation": round(pj["SQ_THREAD_CYCLES_VALU"] / (64.0 * pj["SQ_ACTIVE_INST_VALU"]), 3),
                                 # what share of the issue slots did VALU work on live lanes: VALU share of the instructions x lane utilisation x issue fraction
                                 "useful_lane_frac": round(pj["SQ_INSTS_VALU"] / kernel_s / 1e9 / ISSUE_PEAK_GINST
                                                           * pj["SQ_THREAD_CYCLES_VALU"] / (64.0 * pj["SQ_ACTIVE_INST_VALU"]), 4),
                                 "executed_live_lane_valu": int(executed_live_lane_valu(pj)),
                                 "counters_from": os.path.relpath(PMC_JSON, ROOT), "counters_source_hash": src_hash[:16]})
                    # in-kernel clock under this load: GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / the profiled duration
                    clk = 2.39e9
                    try:
                        clk = pj["GRBM_GUI_ACTIVE"] / 8.0 / (pj["_dispatch"]["duration_ns_under_profiler"] * 1e-9)
                    except Exception:
                        pass
                    mm = mix_model(pj, kernel_s, clk)
                    if mm:
                        roof["mix_model"] = mm
                    hbm.update({"counter_traffic_GBps": round(traffic / kernel_s / 1e9, 1),
                                "counter_traffic_over_peak": round(traffic / kernel_s / 1e9 / HBM_PEAK_GBPS, 4)})
                else:
                    roof["counters_stale"] = (f"{os.path.relpath(PMC_JSON, ROOT)} was captured from another build or workload "
                                              f"(hash {str(pj.get('_source_hash'))[:16]} vs {src_hash[:16]}): not used")
            except Exception as e:      # a malformed profile must not take the benchmark down
                roof["counters_stale"] = f"cannot read {PMC_JSON}: {e}"
        roof["work_model"] = wm
        if "executed_live_lane_valu" in roof:
            # the floor must lie below what the kernel executed on live lanes -- or the table over-prices a unit
            # (recorded, not asserted: a mis-priced unit must not take the headline line down after the timed run -- ADVICE r4;
            #  tests/test_workmodel.py enforces the inequality on the committed counter files)
            wm["floor_over_executed"] = round(wm["lane_slots"] / max(roof["executed_live_lane_valu"], 1), 4)
            wm["floor_violated"] = bool(wm["lane_slots"] > roof["executed_live_lane_valu"])
        out["roofline"] = roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import rzo
        from helpers import oracle_frame, oracle_scene, sync_oracle_flavour
        out["math_flavour"] = sync_oracle_flavour()      # the oracle evaluates sin / cos / acos as the loaded library does (rz_math_flavour())
        ncores = a.cpu_threads if a.cpu_threads > 0 else min(16, len(os.sched_getaffinity(0)))
        osc = oracle_scene(sc)
        ofr = oracle_frame(sc, W, H, a.spp, bounces)
        ref = np.zeros((H, W, 4), np.float32)
        nb = max(1, a.cpu_bands)
        rows, t_cpu, bands, busy = 0, 0.0, [], ncores
        for b in range(nb):
            y0 = min(H - 8, int((b + 0.5) * H / nb) // 8 * 8)
            if y0 in bands:
                continue
            tc = time.perf_counter()
            rzo.render(osc, ofr, accum=ref, crop=(0, y0, W, y0 + 8), nthreads=ncores)   # work is handed out in 16-pixel chunks
            t_cpu += time.perf_counter() - tc
            busy = min(busy, rzo.last_threads_busy())
            rows += 8
            bands.append(y0)
        cpu_samples = rows * W * a.spp
        gpu = final                     # last timed frame (sample_base 0 each step: a complete frame)
        err, same, tot = 0.0, 0, 0
        for y0 in bands:
            g, o = gpu[y0:y0 + 8], ref[y0:y0 + 8]
            err = max(err, float(np.abs(g.astype(np.float64) - o.astype(np.float64)).max()))
            same += int((g.view(np.uint32) == o.view(np.uint32)).all(axis=-1).sum())
            tot += g.shape[0] * g.shape[1]
        cpu_rate = cpu_samples / t_cpu / 1e6
        out["cpu_baseline"] = {"value": round(cpu_rate, 4), "unit": "Msamples/s", "cores": ncores, "threads_busy": busy,
                               "kind": "port",
                               "sample": f"{len(bands)} full-width 8-row bands of the same {W}x{H}x{a.spp}spp frame "
                                         f"({cpu_samples} of {W * H * a.spp} camera paths, {t_cpu:.1f} s wall on {ncores} threads = "
                                         f"{t_cpu * ncores:.0f} core-seconds)",
                               "gpu_over_cpu": round(value / cpu_rate, 1)}
        if a.cpu_full_frame:
            tc = time.perf_counter()
            full = rzo.render(osc, ofr, nthreads=ncores)
            tf = time.perf_counter() - tc
            out["cpu_baseline"]["full_frame"] = {"value": round(W * H * a.spp / tf / 1e6, 4), "seconds": round(tf, 2),
                                                 "bit_identical_to_gpu": bool((full.view(np.uint32) == gpu.view(np.uint32)).all())}
        out["parity"] = {"linf_vs_oracle_on_sample": err, "bit_identical_pixels": same, "pixels_compared": tot}
    if rank == 0 and world > 1 and not a.no_cpu_baseline:
        # N > 1: no CPU baseline (that is the N = 1 line's), but the PARITY leg stays -- a few bands of the frame the group has
        # landed on rank 0, against the oracle at the frame's full sample count: tiles that arrived in the wrong place would
        # pass frame_check (every pixel has the same sample count) and fail here
        from oracle import rzo
        from helpers import oracle_frame, oracle_scene, sync_oracle_flavour
        out["math_flavour"] = sync_oracle_flavour()      # the oracle evaluates sin / cos / acos as the loaded library does (rz_math_flavour())
        ncores = a.cpu_threads if a.cpu_threads > 0 else min(16, len(os.sched_getaffinity(0)))
        osc = oracle_scene(sc)
        ofr = oracle_frame(sc, W, H, spp_total, bounces)
        ref = np.zeros((H, W, 4), np.float32)
        nb = max(1, min(a.cpu_bands, 8))
        bands, same, tot, err = [], 0, 0, 0.0
        for b in range(nb):
            y0 = min(H - 8, int((b + 0.5) * H / nb) // 8 * 8)
            if y0 in bands:
                continue
            rzo.render(osc, ofr, accum=ref, crop=(0, y0, W, y0 + 8), nthreads=ncores)
            bands.append(y0)
            g, o = final[y0:y0 + 8], ref[y0:y0 + 8]
            err = max(err, float(np.abs(g.astype(np.float64) - o.astype(np.float64)).max()))
            same += int((g.view(np.uint32) == o.view(np.uint32)).all(axis=-1).sum())
            tot += g.shape[0] * g.shape[1]
        out["parity"] = {"linf_vs_oracle_on_sample": err, "bit_identical_pixels": same, "pixels_compared": tot,
                         "sample": f"{len(bands)} full-width 8-row bands of the landed {W}x{H}x{spp_total}spp frame (every band crosses the tiles of all {world} ranks)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if group is not None:
        group.close()
    else:
        r.close()
    if mode == "ranks":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
