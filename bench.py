#!/usr/bin/env python3
"""bench.py -- dense carve throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic batch: a fresh model
(all voxels occupied) is carved by all V silhouette views (sphere
of radius 0.35E, ring cameras, 640x480 masks).  The u8 masks and the state plane are
resident in HBM before the timed region starts; everything DERIVED from the masks --
the 1-bit background planes and the summed-area tables the carve kernels read -- is
rebuilt inside every timed step (arvx_set_views_device), then arvx_carve runs.
`value` / `ms_per_step` are that whole step.

`value` / `ms_per_step`: the K steps run ONE AFTER THE OTHER on one context and stream (the
metric is a carve's wall time: reference src/VoxelCarving.cpp:60-72 is one call, one carve), timed
R = 5 times (`rounds`: every round's figure, median, min, max; `value` is the median round), plus one
further round with HIP events around single launches for `carve_kernel_ms` / `views_kernel_ms` / the
`roofline` (the events cost GPU time between dependent kernels: that round is not part of `value`).
`throughput_jobs_in_flight`: the same K steps dealt to --jobs-in-flight contexts (default 4), each on
its own stream -- a caller with a stream of frames hides one job's latency-bound kernels behind
another's issue-bound one; reported beside the headline, never as it.
`e2e`: what SURVEY 8d asks for beside the resident-input figure: page-locked host masks -> device,
the step, and the carved model back on the host -- as two compressed packets (what arvx::Model reads;
`planes_form`: as two plain bit planes).

metric  Mvoxel-views/s = voxels x views / carve time  (BASELINE.json)
N = 1   512^3 grid x 36 views (the configuration the metric is quoted on)
N > 1   one rank per GPU; rank r carves Z slab r of a grid that holds N x 512^3
        voxels (weak scaling), then ONE collective over RCCL merges the
        bit-packed occupancy of all slabs: by default an all-gather of compressed
        packets (two bitmaps + the mixed 64-bit words of each rank's planes, expanded
        by one kernel on the receivers) over striped, load-balanced slabs,
        --collective allgather for the plain in-place all-gather of contiguous slabs,
        --collective allreduce for the north star's all-reduce (striped slabs).
        With several jobs in flight every slot hands its job over in its own stream (pack,
        compress, collective, expand: the other slots' kernels run beside them); with
        --jobs 1 the hand-off of step k runs on a second stream beside step k + 1
        (arvx_ctx_set_exchange_stream).  Every job's hand-off is inside the timed region.

The JSON line also carries `roofline` (algorithmic HBM bytes of SURVEY 8d / the
carve kernel's launch time measured with HIP events on its own stream) and
`cpu_baseline` (the CPU oracle's reference-shaped port, one thread, timed on this
host on a bounded slab of the same workload).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(sc, X, Y, Z, budget_s=12.0):
    """The CPU oracle timed on this host, on a bounded sample of the same workload.
    Primary figure: the reference-shaped port (AoS float voxels, x->y->z loops, M
    recomputed per voxel as the reference does), ONE thread -- the reference is
    single-threaded.  Side figure: the x-fastest OpenMP variant on all cores."""
    from ar_voxel_project_amd import build
    build.build_oracle()
    from oracle import pyoracle
    zc = Z // 2
    t0 = time.perf_counter()
    pyoracle.carve_ref(X, Y, Z, sc.voxel_size, sc.K, sc.Rt, sc.masks, zc, zc + 1)
    t1 = time.perf_counter() - t0  # one plane, to size the sample
    planes = int(max(1, min(Z, budget_s / max(t1, 1e-6))))
    zlo = max(0, zc - planes // 2)
    zhi = min(Z, zlo + planes)
    t0 = time.perf_counter()
    pyoracle.carve_ref(X, Y, Z, sc.voxel_size, sc.K, sc.Rt, sc.masks, zlo, zhi)
    dt = time.perf_counter() - t0
    out = {"value": X * Y * (zhi - zlo) * sc.V / dt / 1e6, "unit": "Mvoxel-views/s",
           "cores": 1, "kind": "port",
           "sample": f"reference-shaped oracle port, planes z={zlo}..{zhi - 1} of "
                     f"{X}x{Y}x{Z} x {sc.V} views ({dt:.1f} s)",
           "published_reference": {"value": 0.71, "unit": "Mvoxel-views/s",
                                   "note": "Report.pdf Fig. 4, i7 @ 4.5 GHz, real reference "
                                           "incl. cv::Mat overhead (BASELINE.md)"}}
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out["cpu_model"] = model
    mt_planes = int(min(Z, max(8, planes * min(ncores, 32) // 4)))
    t0 = time.perf_counter()
    plane = pyoracle.carve(X, Y, mt_planes, sc.voxel_size, sc.M, sc.masks, threads=ncores)
    dmt = time.perf_counter() - t0
    out["all_cores"] = {"value": X * Y * mt_planes * sc.V / dmt / 1e6,
                        "unit": "Mvoxel-views/s", "cores": ncores, "kind": "port",
                        "sample": f"x-fastest OpenMP oracle, planes z=0..{mt_planes - 1} "
                                  f"({dmt:.1f} s)"}
    return out, plane  # plane: the oracle's state of planes z = 0..mt_planes-1


def block_noise_masks(V, H, W, block, seed=0):
    """Masks of random block x block pixel squares, half of them background: every pixel
    rectangle larger than a square is "mixed", so the rectangle tests settle next to nothing --
    the worst case for the culling."""
    rng = np.random.default_rng(seed)
    hb, wb = (H + block - 1) // block, (W + block - 1) // block
    coarse = rng.random((V, hb, wb)) >= 0.5
    m = np.repeat(np.repeat(coarse, block, axis=1), block, axis=2)[:, :H, :W]
    return (m * 255).astype(np.uint8)


def sparse_noise_masks(V, H, W, block, p_bg, seed=0):
    """Mostly foreground with a fraction p_bg of block x block background squares: no rectangle test
    settles anything (every rectangle holds both kinds) and nearly every voxel stays occupied through
    all views -- the regime the culling cannot help and the early-outs do not end."""
    rng = np.random.default_rng(seed)
    hb, wb = (H + block - 1) // block, (W + block - 1) // block
    coarse = rng.random((V, hb, wb)) >= p_bg
    m = np.repeat(np.repeat(coarse, block, axis=1), block, axis=2)[:, :H, :W]
    return (m * 255).astype(np.uint8)


def run_workloads(capi, synthetic, dev, check):
    """The other configurations behind the headline, each in a few launches: the BASELINE.json
    configurations that fit one GPU, a worst case for the culling, and the north-star target
    with its planes checked.  Per entry: carve_kernel_ms (HIP events around arvx_carve, best of
    5), the fraction of (sub-tile, view) pairs the rectangle tests could NOT settle
    (ARVX_CARVE_STATS, an untimed extra launch) and -- `check`: the CPU oracle as checker --
    parity of the state with the oracle, every voxel of the planes named."""
    from tests import golden_io  # the reference's own silhouettes / photographs (fixtures)
    stream = torch.cuda.current_stream()
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = []

    def carve_entry(name, N, M, masks, s, planes=None, post=None, no_cull_too=False):
        V = masks.shape[0]
        e = {"workload": name, "grid": [N, N, N], "views": V}
        with capi.Context(N, N, N, s, device=dev.index) as ctx:
            ctx.set_stream(stream.cuda_stream)
            ctx.set_views(M, masks, campos=post["campos"] if post else None)
            best = 1e9
            for _ in range(5):
                ctx.reset()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                ctx.carve(0)
                b.record(stream)
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            e["carve_kernel_ms"] = best
            e["value"] = N ** 3 * V / best / 1e3
            e["unit"] = "Mvoxel-views/s"
            if no_cull_too:  # the brute-force kernel on the same input, for comparison
                bf = 1e9
                for _ in range(3):
                    ctx.reset()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    ctx.carve(capi.CARVE_NO_CULL)
                    b.record(stream)
                    torch.cuda.synchronize()
                    bf = min(bf, a.elapsed_time(b))
                e["no_cull_kernel_ms"] = bf
                ctx.reset()
                ctx.carve(0)
            st = ctx.download_state() if (check or post) else None
            if st is not None:
                e["occupied_fraction"] = float((st & 1).mean())
            if post:  # carve -> colour vote -> handleUnseen -> closure -> mesh, src/main.cpp:262-303
                ctx.set_images(post["images"])
                # the stages after the carve, four rounds of the whole sequence on the same context:
                # the first pays the first uses (pool allocations), the others are warm
                per_round = []
                for rnd in range(4):
                    if rnd:
                        ctx.reset()
                        ctx.carve(0)
                    stages = {}
                    for stage, call in (("colour_avg", lambda: ctx.color(capi.COLOR_AVERAGE)),
                                        ("handle_unseen", ctx.handle_unseen),
                                        ("closure_3", lambda: ctx.closure(3, True, download=False)),
                                        ("mc_mesh", lambda: ctx.mc_mesh_count(True))):
                        ctx.synchronize()
                        t0 = time.perf_counter()
                        r = call()
                        ctx.synchronize()
                        stages[stage] = (time.perf_counter() - t0) * 1e3
                        if stage == "mc_mesh":
                            e["triangles"] = int(r)
                    per_round.append(stages)
                e["stage_ms"] = {k: min(pr[k] for pr in per_round[1:]) for k in per_round[0]}
                e["stage_ms_first_call"] = per_round[0]
                e["stage_ms_note"] = ("wall time of the C-ABI call with inputs resident, incl. its "
                                      "own synchronisations; no result copied to the host; "
                                      "stage_ms = best of 3 warm rounds, stage_ms_first_call = the "
                                      "cold first round (pool allocations)")
            ctx.reset()
            ctx.carve(capi.CARVE_STATS)
            stt = ctx.stats()
            # pairs that needed per-voxel work, of those the sub-tile stage looked at (the rest was
            # settled per coarse tile before) and of all (sub-tile, view) pairs of the grid
            e["mixed_pair_fraction"] = stt["subtile_views_mixed"] / max(1, stt["subtile_views_total"])
            all_pairs = ((N + 15) // 16) * ((N + 7) // 8) ** 2 * V
            e["mixed_pairs_of_all"] = stt["subtile_views_mixed"] / all_pairs
            e["subtiles_carved_by_a_rectangle"] = stt["subtiles_carved"] / max(1, stt["subtiles"])
        if check:
            from oracle import pyoracle
            if planes is None:
                want = pyoracle.carve(N, N, N, s, M, masks, threads=ncores)
                got = st
                e["parity_planes"] = f"all {N}"
            else:
                want = pyoracle.carve_planes(N, N, s, M, masks, planes, threads=ncores)
                got = st[planes]
                e["parity_planes"] = f"z={planes[0]}..{planes[-1]}"
            e["parity_vs_oracle"] = bool(np.array_equal(got, want))
        return e

    def guarded(fn):
        try:
            out.append(fn())
        except Exception as ex:  # noqa: BLE001 -- a workload that cannot run must not cost the line
            out.append({"error": f"{type(ex).__name__}: {ex}"})

    sc = synthetic.sphere_scene(128, 8)
    guarded(lambda: carve_entry("C1 Data/box_dataset silhouettes (8 views, PIL-decoded, ring cameras), 128^3",
                                128, sc.M, golden_io.dataset_masks("box"), sc.voxel_size))
    sc = synthetic.sphere_scene(256, 24)
    human = golden_io.dataset_masks("human", recentre=True)
    guarded(lambda: carve_entry("C2 Data/human_dataset silhouettes (24 views, re-centred, ring cameras), 256^3",
                                256, sc.M, human, sc.voxel_size))
    sc5 = synthetic.sphere_scene(512, 24)
    guarded(lambda: carve_entry("C5 Data/human_dataset 512^3 x 24: carve + colour vote + handleUnseen + "
                                "closure + marching-cubes mesh", 512, sc5.M, human, sc5.voxel_size,
                                post={"campos": sc5.campos, "images": golden_io.dataset_images("human")}))
    sc = synthetic.sphere_scene(512, 36)
    guarded(lambda: carve_entry("ragged silhouettes: 512^3 x 36 views of 8x8-pixel block noise",
                                512, sc.M, block_noise_masks(36, sc.H, sc.W, 8), sc.voxel_size))
    guarded(lambda: carve_entry("worst case for the culling: 512^3 x 36 views of 2x2-pixel block noise "
                                "(no pixel rectangle of a 4x4x4 block is uniform)",
                                512, sc.M, block_noise_masks(36, sc.H, sc.W, 2), sc.voxel_size))
    # what neither the rectangles nor the early-outs can help: sparse background
    for name, blk, pbg in (("2x2-pixel noise, 2 % background", 2, 0.02),
                           ("2x2-pixel noise, 10 % background", 2, 0.10),
                           ("all foreground with 0.5 % isolated background pixels", 1, 0.005)):
        guarded(lambda name=name, blk=blk, pbg=pbg: carve_entry(
            f"no rectangle settles, voxels stay alive: 512^3 x 36 views, {name}", 512, sc.M,
            sparse_noise_masks(36, sc.H, sc.W, blk, pbg), sc.voxel_size,
            planes=np.arange(192, 320), no_cull_too=True))
    sc = synthetic.sphere_scene(1024, 36)
    guarded(lambda: carve_entry("north-star target: sphere, 1024^3 x 36 views", 1024, sc.M, sc.masks,
                                sc.voxel_size, planes=np.arange(384, 640)))
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run
    --nproc-per-node N bench.py <the same arguments>` as a child process (one rank per GPU, RCCL
    over 127.0.0.1), pass its output through, and check that the line rank 0 printed speaks of N
    GPUs.  -> the exit status for this process (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           *sys.argv[1:]]
    print(f"bench.py: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for text in child.stdout:
        sys.stdout.write(text)
        sys.stdout.flush()
        if text.startswith("{"):
            line = text
    rc = child.wait()
    if rc != 0:
        print(f"bench.py: the ranks exited with status {rc}", file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    try:
        got = json.loads(line)["n_gpus"]
    except (TypeError, ValueError, KeyError):
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    if got != n:
        print(f"bench.py: --gpus {n} but the line says n_gpus = {got}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, default=512, help="base grid edge per GPU")
    ap.add_argument("--views", type=int, default=36)
    ap.add_argument("--no-cull", action="store_true", help="evaluate every voxel in every view")
    ap.add_argument("--collective", default="compressed",
                    choices=["allreduce", "allgather", "compressed", "packets", "none"],
                    help="end-of-carve merge of the packed occupancy: all-gather of compressed "
                         "packets (default: bitmaps + mixed words, ~1/7 of the bytes; checked "
                         "by an untimed trial first and replaced by the plain all-gather if the "
                         "trial fails or a packet overflows), in-place all-gather of contiguous "
                         "slabs, or the north star's all-reduce (SUM over zero-filled planes, "
                         "striped slabs); `packets`: the compressed all-gather whose product STAYS the "
                         "ranks' packets (each with its own index) -- the plain merged plane is built "
                         "on demand, outside the step (reported beside the others under `collectives`)")
    ap.add_argument("--jobs", type=int, default=1,
                    help="jobs in flight for the HEADLINE (default 1: one step after the other, what "
                         "the metric means); > 1 deals the steps to that many contexts and streams")
    ap.add_argument("--jobs-in-flight", type=int, default=4,
                    help="the `throughput_jobs_in_flight` leg: every job is a whole step (fresh "
                         "model, views derived, carved); job k runs on context and stream k %% jobs, "
                         "so that the latency-bound head of one job runs beside the issue-bound "
                         "exact kernel of another (0 = skip the leg)")
    ap.add_argument("--rounds", type=int, default=5,
                    help="the timed region (K steps) is run this many times: median / min / max")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (PCIe-inclusive) leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-ablation", action="store_true",
                    help="skip the NO_CULL ablation leg (keeps a profile to one kernel variant)")
    ap.add_argument("--extra-grid", type=int, default=1024,
                    help="also time this grid at N=1 (0 = skip); reported under 'extra'")
    ap.add_argument("--no-mgpu", action="store_true",
                    help="N > 1: skip the one-process-all-GPUs leg (libarvx_mgpu.so)")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the `workloads` array (the other BASELINE configurations, a worst "
                         "case for the culling and the 1024^3 target with parity)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: this process starts the N ranks itself --
        # before anything here has touched a GPU -- as fresh child processes, relays their output
        # and exits with their status.  (A launcher that sets WORLD_SIZE itself goes on below.)
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never measure another number of GPUs than the line will say
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # ARVX_BENCH_ONE_GPU=1 rehearses the N>1 code path on a one-GPU box (all ranks
    # on cuda:0, gloo instead of RCCL); never set by the driver.
    rehearsal = os.environ.get("ARVX_BENCH_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ar_voxel_project_amd import build, capi, sharding, synthetic
    if not os.path.exists(capi.LIB_PATH):  # a checkout without built artefacts
        if local_rank == 0:
            build.build_library()
        if world > 1:
            dist.barrier()

    def barrier():
        if world > 1:
            dist.barrier()

    # HIP events inside the timed region: every event is a marker packet between two dependent
    # kernels, and three of them per step cost 13.6 us of a 0.146 ms step (measured: all /
    # carve only / none = 0.1456 / 0.1402 / 0.1320 ms per step).  They are therefore recorded on
    # a sample of the timed steps: five of them (fewer when K < 20), evenly spaced.
    EV = {"none": 0, "carve": 1, "all": 2}[os.environ.get("ARVX_BENCH_EVENTS", "all")]

    def run_config(base, V, steps, warmup, collective, no_cull=False, jobs=1, rounds=1):
        jobs = max(1, jobs)
        rounds = max(1, rounds)
        X, Y, Z = sharding.grid_for(world, base)
        sc = synthetic.sphere_scene(max(X, Y, Z), V)
        sc.X, sc.Y, sc.Z = X, Y, Z
        zlo, zhi = sharding.slab_of(Z, world, rank)
        nvox_global = X * Y * Z
        flags = capi.CARVE_NO_CULL if (args.no_cull or no_cull) else 0
        # striped (load-balanced) slabs wherever the collective can place scattered words;
        # the plain all-gather needs contiguous ones
        layout = "striped" if (world > 1 and collective != "allgather") else "slab"
        # one context and one real (non-null) HIP stream per job slot, shared by torch and the
        # library, so that the torch.cuda.Event pairs below bracket exactly the carve launch
        ctxs, streams = [], []
        for _ in range(jobs):
            if layout == "striped":
                c = capi.Context(X, Y, Z, sc.voxel_size, device=local_rank, stripes=(world, rank))
            else:
                c = capi.Context(X, Y, Z, sc.voxel_size, device=local_rank, z_range=(zlo, zhi))
            st = torch.cuda.Stream(device=dev)
            c.set_stream(st.cuda_stream)
            ctxs.append(c)
            streams.append(st)
        ctx, stream = ctxs[0], streams[0]
        torch.cuda.set_stream(stream)
        d_masks = torch.from_numpy(sc.masks).to(dev)  # resident before timing
        torch.cuda.synchronize()  # (the upload ran on the null stream)
        for c in ctxs:  # set-up of a slot: its buffers exist before anything is counted
            c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
            if jobs > 1:  # (the W warm-up steps may not reach every slot)
                c.reset()
                c.carve(flags)
        torch.cuda.synchronize()
        # N > 1: the merge of the packed occupancy.  One job at a time: ONE exchange object with
        # two packed buffers and the hand-off on a side stream (below).  Several jobs in flight:
        # every slot has an exchange object of its own and hands its job over in its OWN stream
        # -- no event between streams at all; the other slots' kernels run beside it
        # (tools/rank_step_time.py: a rank's step at 8 GPUs without the collective, 4 jobs in
        # flight: 91 us with two side streams and their events, 88 us this way; 134 us for one
        # job at a time with the hand-off behind the carve, 120 us beside the next one).
        own = jobs > 1
        ex, exs = None, []
        lazy = collective == "packets"  # the compressed exchange, its product kept as packets
        if lazy:
            collective = "compressed"
        if world > 1 and collective != "none":
            exs = [sharding.OccupancyExchange(X, Y, Z, world, rank, dev, mode=collective,
                                              buffers=1 if own else 2, layout=layout, codec=ctxs[k],
                                              lazy_expand=lazy)
                   for k in range(jobs if own else 1)]
            ex = exs[0]
        n_ev = max(1, min(5, steps // 4))  # timed steps that carry events, evenly spaced
        ev = {(2 * k + 1) * steps // (2 * n_ev): tuple(torch.cuda.Event(enable_timing=True)
                                                       for _ in range(3))
              for k in range(n_ev)}
        if jobs > 1:  # kernels of several jobs interleave: a launch has no duration of its own
            ev = {}
        nstep = [0]
        merge_ok = [None]
        # N > 1: the hand-off of job k -- pack, compress, the collective, expand: HBM-bound
        # kernels, 78 us of a rank's 176 us at 8 GPUs when they run behind the carve
        # (tools/rank_step_time.py) -- goes to a stream of its own and runs beside the views
        # and the carve of job k + 1.  Two events order the streams: the pack waits for the
        # carve whose records it reads, the next carve for that pack.
        # (Several jobs in flight: see above; the collectives are issued in job order, the same
        # order on every rank.)
        sides = ([torch.cuda.Stream(device=dev) for _ in range(2)]
                 if ex is not None and not own else None)
        packed = [None] * jobs  # per slot: recorded behind the slot's latest pack

        def step(i=None):
            # one job: fresh model -> carve all views -> (N>1) merge the occupancy of
            # all slabs.  The collective of job k runs on RCCL's stream while job
            # k+1 carves (two packed buffers); the final wait is inside the timing.
            if i not in ev:
                i = None
            slot = nstep[0] % jobs
            ctx, stream = ctxs[slot], streams[slot]
            ctx.reset()
            if i is not None and EV >= 2:
                ev[i][0].record(stream)
            # masks are the resident input; bit planes + summed-area tables are derived
            ctx.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
            if i is not None and EV >= 1:
                ev[i][1].record(stream)
            if packed[slot] is not None:
                stream.wait_event(packed[slot])  # the slot's previous pack has read the records
            ctx.carve(flags)
            if i is not None and EV >= 1:
                ev[i][2].record(stream)
            if ex is not None and own:
                xs = exs[slot]
                with torch.cuda.stream(stream):  # (the collective is ordered against this stream)
                    xs.prepare(0, verify=False)  # the slot's previous exchange: wait, expand
                    xs.pack(ctx, 0)  # compressed: records -> packet (own words into the plane)
                    xs.launch(0, async_op=True)
            elif ex is not None:
                b = nstep[0] % 2
                side = sides[b]
                carved = torch.cuda.Event()
                carved.record(stream)
                ctx.set_exchange_stream(side.cuda_stream)  # pack / compress / expand of this job
                ex.codec = ctx
                with torch.cuda.stream(side):
                    side.wait_event(carved)
                    ex.prepare(b, verify=False)  # compressed: overflow is checked after drain()
                    ex.pack(ctx, b)  # compressed: records -> packet (own words into the plane)
                    packed[slot] = torch.cuda.Event()
                    packed[slot].record(side)
                    ex.launch(b, async_op=True)
            nstep[0] += 1

        def drain():
            if ex is not None and own:
                for k, xs in enumerate(exs):
                    with torch.cuda.stream(streams[k]):
                        xs.wait(0, verify=False)
            elif ex is not None:
                for b in range(2):
                    ctxs[0].set_exchange_stream(sides[b].cuda_stream)  # (the expand of buffer b)
                    ex.codec = ctxs[0]
                    with torch.cuda.stream(sides[b]):
                        ex.wait(b, verify=False)
            torch.cuda.synchronize()

        # (the interpreter's cyclic collector pauses for 40-85 ms once torch is loaded -- seen at
        # the 543rd set_views_device call, tools/host_probe.py -- and a pause that long drains
        # the launch queue: no collections inside the timed region.  Collected BEFORE the warm-up
        # steps: the collection itself leaves the GPU idle for tens of milliseconds, and the
        # steps that follow an idle period run slower)
        gc.collect()
        gc.disable()
        for _ in range(warmup):
            step()
        drain()
        packet_cap = None
        if ex is not None and collective == "compressed":
            if nstep[0] == 0:  # --warmup 0: one untimed job to size the packets
                step()
                drain()
            # packets start at the worst-case size; size them to what this scene needs
            if own:  # (from the slot of the latest job; the same packets on every rank)
                packet_cap = exs[(nstep[0] - 1) % jobs].retune(0)
                for xs in exs:
                    xs.cap = packet_cap
            else:
                packet_cap = ex.retune((nstep[0] - 1) % 2)
        # `rounds` timed regions of exactly `steps` steps each, bracketed by barrier + synchronize
        # on both sides, without events; then (one job at a time) one more region WITH the events
        # for the kernel times -- not part of the round figures
        dts = []
        try:
            for rnd in range(rounds + (1 if ev else 0)):
                with_ev = rnd == rounds
                barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    step(i if with_ev else None)
                drain()
                barrier()
                torch.cuda.synchronize()
                if not with_ev:
                    dts.append(time.perf_counter() - t0)
        finally:
            gc.enable()
        tmax = torch.tensor(dts, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)  # per round: the slowest rank
        dts = [float(x) for x in tmax.tolist()]
        dt = float(np.median(dts))
        views_ms = (float(np.mean([a.elapsed_time(b) for a, b, _ in ev.values()]))
                    if EV >= 2 and ev else float("nan"))
        kern_ms = (float(np.mean([b.elapsed_time(c) for _, b, c in ev.values()]))
                   if EV >= 1 and ev else float("nan"))
        occ = None
        overflowed = any(xs.overflowed() for xs in exs)
        # the slot of the LAST job: its planes are what the last exchange merged
        last = (nstep[0] - 1) % jobs
        merged_plane = None
        if ex is not None:  # (`packets`: expanded here, on demand -- after the timed region)
            for c in ctxs:
                c.set_exchange_stream(0)
            xs_last = exs[last] if own else ex
            xs_last.codec = ctxs[last] if own else ctxs[0]
            merged_plane = xs_last.merged_plane(0 if own else (nstep[0] - 1) % 2, verify=False)
            torch.cuda.synchronize()
        ctx = ctxs[last]
        st = ctx.download_state() if (rank == 0 or ex is not None) else None
        # every slot that ran a job must hold the same model
        slots_agree = True
        if st is not None:
            for k, c in enumerate(ctxs):
                if k != last and k < nstep[0]:
                    slots_agree = slots_agree and bool(np.array_equal(c.download_state(), st))
        merged_count_ok = None
        if ex is not None:  # the merged plane must hold as many voxels as the slabs together
            cnt = torch.tensor([int((st & 1).sum())], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            if rank == 0:
                merged = merged_plane.cpu().numpy().view(np.uint64)
                merged_count_ok = bool(int(np.bitwise_count(merged).sum()) == int(cnt.item()))
        if rank == 0:
            occ = float((st & 1).mean())
            if ex is not None:  # the merged plane must hold this rank's planes
                got = merged_plane.cpu().numpy().view(np.uint8)
                got = got.reshape(Z, X * Y // 8)[ctx.planes]
                mine = np.packbits((st.reshape(len(ctx.planes), -1) & 1).astype(np.uint8),
                                   axis=1, bitorder="little")
                merge_ok[0] = bool(np.array_equal(got, mine)) and merged_count_ok
        for c in ctxs:
            c.close()
        del d_masks
        nplanes = len(ctx.planes) if hasattr(ctx, "planes") else zhi - zlo
        xbytes = None
        if ex is not None:  # bytes each rank contributes to the collective
            xbytes = ((ex.header + packet_cap) * 8 if collective == "compressed"
                      else ex.total_words * 4 if collective == "allreduce" else ex.my_words * 4)
        kern_ms = None if kern_ms != kern_ms else kern_ms  # (no NaN in the JSON line)
        views_ms = None if views_ms != views_ms else views_ms
        return dict(X=X, Y=Y, Z=Z, V=V, dt=dt, dts=dts, kern_ms=kern_ms, views_ms=views_ms, sc=sc, occ=occ,
                    state=st if (world == 1 and rank == 0) else None,
                    nplanes=nplanes, layout=layout, ev_steps=len(ev), nvox=nvox_global, merge_ok=merge_ok[0],
                    jobs=jobs, slots_agree=slots_agree,
                    overflowed=overflowed, exchange_bytes_per_rank=xbytes)

    if world > 1 and args.collective == "compressed":
        # untimed trial of the compressed exchange: every rank must see a merged plane
        # that holds all slabs' voxels, else all ranks use the plain all-gather
        try:
            t = run_config(args.grid, args.views, 1, 1, "compressed")
            good = (t["merge_ok"] is not False) and not t["overflowed"]
        except Exception as e:  # noqa: BLE001 -- any failure means "do not use it"
            print(f"[bench] rank {rank}: compressed exchange trial failed: {e}", file=sys.stderr)
            good = False
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()):
            args.collective = "allgather"
    def measure(base, V, steps, warmup, collective, jobs, rounds=1):
        """K steps, `jobs` in flight (1: one after the other), `rounds` times.  With jobs > 1 a launch
        has no duration of its own: the kernel times then come from a short run of one job after
        the other in the same frame."""
        m = run_config(base, V, steps, warmup, collective, jobs=jobs, rounds=rounds)
        m["latency_ms"] = m["dt"] / steps * 1e3
        m["latency_steps"] = steps
        if jobs > 1 and not m["overflowed"]:
            k = max(8, steps // 2)
            q = run_config(base, V, k, min(warmup, 5), collective, jobs=1)
            m.update(kern_ms=q["kern_ms"], views_ms=q["views_ms"], ev_steps=q["ev_steps"],
                     latency_ms=q["dt"] / k * 1e3, latency_steps=k)
        return m

    r = measure(args.grid, args.views, args.steps, args.warmup, args.collective, args.jobs, args.rounds)
    if r["overflowed"]:
        # a packet outgrew its size inside the timed region: that run does not count
        args.collective = "allgather"
        r = measure(args.grid, args.views, args.steps, args.warmup, args.collective, args.jobs, args.rounds)
    vv = r["nvox"] * r["V"]
    value = vv * args.steps / r["dt"] / 1e6
    ms_per_step = r["dt"] / args.steps * 1e3
    per_round = [d / args.steps * 1e3 for d in r["dts"]]
    rounds_out = {"n": len(per_round), "ms_per_step": per_round, "median": float(np.median(per_round)),
                  "min": min(per_round), "max": max(per_round),
                  "spread": (max(per_round) - min(per_round)) / float(np.median(per_round)),
                  "note": f"every round = exactly {args.steps} steps one after the other between "
                          "barrier + synchronize; `value` / `ms_per_step` = the median round"}

    # roofline of the carve kernels: SURVEY 8(d) algorithmic bytes, HBM-read side:
    # N*V (one state byte per voxel-view) + V*W*H (one mask byte per pixel), for the
    # voxels THIS rank's launch processes.
    nv_rank = r["X"] * r["Y"] * r["nplanes"]
    alg_bytes = nv_rank * r["V"] + r["V"] * r["sc"].W * r["sc"].H
    kern_ms = r["kern_ms"]  # None without events (ARVX_BENCH_EVENTS=none)
    views_ms = r["views_ms"]
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms else None
    traffic = traffic_step = None
    valu = valu_all = valu_views = None
    profile_note = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            key = f"{r['X']}x{r['Y']}x{r['Z']}x{r['V']}" + ("_nocull" if args.no_cull else "")
            ent = tj.get(key, {})
            stamp = build.source_stamp()
            if ent and ent.get("source_stamp") != stamp:
                # the PMC pass was taken on other kernel sources: its counts say nothing about this build
                profile_note = (f"profiles/traffic.json[{key}] was taken on kernel sources "
                                f"{ent.get('source_stamp')}, this build is {stamp}: no PMC figures")
                ent = {}
            traffic = ent.get("bytes_per_launch")
            traffic_step = ent.get("bytes_per_step")
            valu_all = ent.get("valu_wave_instructions") or None
            valu_views = ent.get("valu_wave_instructions_views") or None
            ins = ent.get("valu_wave_instructions", {}).get("exact")
            t_ns = ent.get("kernel_avg_ns_rocprofv3", {}).get("exact")
            if ins and t_ns:
                peak = 256 * 4 * 2.4e9 / 2
                valu = {"kernel": "carve_exact_blocks_kernel", "SQ_INSTS_VALU": ins,
                        "kernel_us_rocprofv3": t_ns / 1e3, "achieved_Ginst_per_s": ins / t_ns,
                        "peak_Ginst_per_s": peak / 1e9, "frac": ins / (t_ns * 1e-9) / peak,
                        "source": "profiles/traffic.json (PMC pass of the same command, same sources)"}
        except Exception as ex:  # noqa: BLE001
            profile_note = f"profiles/traffic.json unreadable: {ex}"
            traffic = None
    phys = (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms) else None
    step_kernel_ms = (kern_ms + views_ms) if (kern_ms and views_ms) else None
    # What bounds the carve is vector-instruction issue (the exact kernel) and dependent-read
    # latency (the classification), not HBM: the roofline is stated against the issue peak --
    # 256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles at 2.4 GHz (the guide's FP32 vector
    # peak counted in instructions; fp64 adds / converts issue at half that).  Work = the VALU
    # wave-instructions of the launch's three kernels (SQ_INSTS_VALU, a PMC pass of the same
    # command: profiles/traffic.json -- deterministic for a fixed scene), time = this run's.
    VALU_PEAK = 256 * 4 * 2.4e9 / 2 / 1e9  # G wave-instructions / s
    ins_all = sum(valu_all.values()) if valu_all else None
    ok_k = bool(ins_all and kern_ms)
    roofline = {"bound": "valu-issue",
                "achieved": (ins_all / (kern_ms * 1e-3) / 1e9) if ok_k else None,
                "peak": VALU_PEAK, "unit": "G wave-instructions/s",
                "frac": (ins_all / (kern_ms * 1e-3) / 1e9 / VALU_PEAK) if ok_k else None,
                "traffic": traffic,
                "kernel": "carve_coarse_kernel + carve_classify_dense_kernel + "
                          "carve_exact_blocks_kernel<LEFT> (one arvx_carve call of a fresh model)",
                "kernel_ms": kern_ms,
                "valu_wave_instructions": valu_all,
                "dominant_kernel": valu,
                "profile_note": profile_note,
                "measured_here": "kernel_ms (HIP events on the launch stream)",
                "from_profiles": "valu_wave_instructions, traffic, dominant_kernel: "
                                 "profiles/traffic.json (rocprofv3 --pmc passes of this command, tied "
                                 "to the kernel sources by `source_stamp`)",
                "hbm_effective": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                                  "algorithmic_bytes": alg_bytes,
                                  "note": "SURVEY 8d's per-view streaming formulation (N*V + V*W*H "
                                          "read bytes) / kernel time: an EFFECTIVE rate -- the "
                                          "kernels keep 2 bits per voxel, write them once and settle "
                                          "~99 % of the (sub-tile, view) pairs from pixel "
                                          "rectangles, so it exceeds the physical peak"},
                "hbm_physical": {"traffic_bytes": traffic, "GBps": phys,
                                 "frac": (phys / HBM_PEAK_GBS) if phys else None,
                                 "note": "2 x FETCH_SIZE + WRITE_SIZE per the guide's gfx950 "
                                         "correction"},
                "step": {"kernels": "views_strip_kernel (arvx_set_views_device) + the three carve kernels",
                         "kernel_ms": step_kernel_ms,
                         "valu_wave_instructions": ((ins_all + sum(valu_views.values()))
                                                    if (ins_all and valu_views) else None),
                         "hbm_effective_frac": (alg_bytes / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                         if step_kernel_ms else None,
                         "traffic": traffic_step,
                         "hbm_physical_frac": (traffic_step / (step_kernel_ms * 1e-3) / 1e9 /
                                               HBM_PEAK_GBS) if (traffic_step and step_kernel_ms) else None}}

    out = {
        "metric": "Mvoxel-views/s (voxels x views / s) + carve wall-time, 512^3 grid x 36 views",
        "value": value, "unit": "Mvoxel-views/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"synthetic sphere silhouettes, {r['X']}x{r['Y']}x{r['Z']} grid, "
                               f"{r['V']} views 640x480, per step: derive bit planes + summed-area "
                               f"tables from the resident u8 masks, then dense carve of a fresh "
                               f"model by all views; one step after the other",
                   "grid": [r["X"], r["Y"], r["Z"]], "views": r["V"],
                   "jobs_in_flight": r["jobs"],
                   "parallelism": f"z-slab x{world} ({r['layout']})" if world > 1 else "single GPU",
                   "collective": args.collective if world > 1 else "none",
                   "merged_plane_holds_rank0_planes": r["merge_ok"],
                   "exchange_bytes_per_rank": r["exchange_bytes_per_rank"],
                   "cull": not args.no_cull},
        "rounds": rounds_out,
        "latency_ms_per_step": r["latency_ms"],
        "slots_agree": r["slots_agree"],
        "carve_kernel_ms": kern_ms, "views_kernel_ms": views_ms,
        "kernel_ms_from": f"HIP events on the launch stream around {r['ev_steps']} of the "
                          f"{r.get('latency_steps', args.steps)} steps of one further round (an "
                          f"event is a packet between dependent kernels; that round is not part of "
                          f"`value`)",
        "carve_only_value": (vv / (kern_ms * 1e-3) / 1e6) if kern_ms else None,
        "occupied_fraction": r["occ"],
        "roofline": roofline,
    }

    if world == 1 and rank == 0 and args.jobs_in_flight > 1 and args.jobs == 1:
        # the same steps with several jobs in flight: a throughput figure of its own, not the headline
        try:
            tj_ = run_config(args.grid, args.views, args.steps, min(args.warmup, 5), "none",
                             jobs=args.jobs_in_flight, rounds=min(3, args.rounds))
            out["throughput_jobs_in_flight"] = {
                "jobs": args.jobs_in_flight, "value": vv * args.steps / tj_["dt"] / 1e6,
                "unit": "Mvoxel-views/s", "ms_per_step": tj_["dt"] / args.steps * 1e3,
                "rounds_ms_per_step": [d / args.steps * 1e3 for d in tj_["dts"]],
                "slots_agree": tj_["slots_agree"],
                "note": f"{args.steps} whole steps dealt to {args.jobs_in_flight} contexts, each on its "
                        "own stream (job k on slot k % jobs); every slot's final model compared"}
        except Exception as ex:  # noqa: BLE001
            out["throughput_jobs_in_flight"] = {"error": f"{type(ex).__name__}: {ex}"}

    if world > 1:
        # One scaling run answers every collective question: the headline above is the default
        # (compressed all-gather over striped slabs); the north star's all-reduce and the plain
        # all-gather follow with a few steps each -- the same job, the same timing frame, not
        # part of `value`.
        def coll_entry(c, k):
            xb = c["exchange_bytes_per_rank"]
            return {"ms_per_step": c["dt"] / k * 1e3, "value": c["nvox"] * c["V"] / (c["dt"] / k) / 1e6,
                    "unit": "Mvoxel-views/s", "steps": k, "layout": c["layout"],
                    "jobs_in_flight": c["jobs"], "latency_ms_per_step": c["latency_ms"],
                    "exchange_bytes_per_rank": xb, "merge_ok": c["merge_ok"],
                    "carve_kernel_ms": c["kern_ms"], "views_kernel_ms": c["views_ms"]}
        coll = {args.collective: coll_entry(r, args.steps)}
        kc = max(5, args.steps // 10)
        for mode in ("compressed", "allreduce", "allgather", "packets"):
            if mode in coll:
                continue
            try:
                coll[mode] = coll_entry(measure(args.grid, args.views, kc, 2, mode, args.jobs), kc)
            except Exception as ex:  # noqa: BLE001
                coll[mode] = {"error": f"{type(ex).__name__}: {ex}"}
        if rank == 0:
            out["collectives"] = coll
            out["collective_backend"] = {
                "backend": dist.get_backend(), "ranks": dist.get_world_size(),
                "library": ("gloo (one-GPU rehearsal)" if rehearsal else
                            "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version()))}

    if world == 8 and (args.grid, args.views) == (512, 36):
        # BASELINE config 4 as it is written: 1024^3 x 72 views over the 8 GPUs
        try:
            k = max(3, args.steps // 2)
            c4 = measure(512, 72, k, 2, args.collective, args.jobs)
            if rank == 0:
                out["c4_1024x72"] = {
                    "workload": f"{c4['X']}x{c4['Y']}x{c4['Z']} x 72 views, 8 GPUs",
                    "value": c4["nvox"] * 72 / (c4["dt"] / k) / 1e6, "unit": "Mvoxel-views/s",
                    "ms_per_step": c4["dt"] / k * 1e3, "latency_ms_per_step": c4["latency_ms"],
                    "jobs_in_flight": c4["jobs"], "carve_kernel_ms": c4["kern_ms"],
                    "views_kernel_ms": c4["views_ms"], "merge_ok": c4["merge_ok"]}
        except Exception as ex:  # noqa: BLE001
            if rank == 0:
                out["c4_1024x72"] = {"error": str(ex)}

    if world > 1 and not rehearsal and not args.no_mgpu:
        # The same job driven by ONE process over all GPUs (libarvx_mgpu.so: ncclCommInitAll, one
        # stream per device) -- the form a drop-in caller of the reference's carve() uses
        # (include/arvx/multi_gpu.hpp).  Run as a child process of rank 0 while the ranks of this
        # job wait on the host (a gloo barrier: their GPUs are idle), with a time limit: a
        # failure there must not cost the line.
        host = dist.new_group(backend="gloo")
        if rank == 0:
            import subprocess
            try:
                cp = subprocess.run([sys.executable, "-m", "ar_voxel_project_amd.mgpu", "--devices",
                                     str(world), "--grid", str(args.grid), "--views", str(args.views),
                                     "--steps", str(max(5, args.steps // 10))],
                                    cwd=ROOT, capture_output=True, text=True, timeout=240)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                out["arvx_mgpu"] = (json.loads(line[-1]) if cp.returncode == 0 and line else
                                    {"error": f"rc {cp.returncode}: {cp.stderr[-400:]}"})
            except Exception as ex:  # noqa: BLE001 (incl. TimeoutExpired)
                out["arvx_mgpu"] = {"error": f"{type(ex).__name__}: {ex}"}
        dist.barrier(group=host)

    if rank == 0 and world == 1 and args.extra_grid and args.extra_grid != args.grid:
        try:
            k = max(3, args.steps // 4)
            e = measure(args.extra_grid, args.views, k, 1, "none", args.jobs, min(3, args.rounds))
            evv = e["nvox"] * e["V"]
            eb = evv + e["V"] * e["sc"].W * e["sc"].H
            out["extra"] = {
                "workload": f"{e['X']}^3 x {e['V']} views (north-star target config)",
                "value": evv / (e["dt"] / k) / 1e6,
                "unit": "Mvoxel-views/s", "carve_kernel_ms": e["kern_ms"],
                "views_kernel_ms": e["views_ms"], "ms_per_step": e["dt"] / k * 1e3,
                "jobs_in_flight": e["jobs"], "latency_ms_per_step": e["latency_ms"],
                "rounds_ms_per_step": [d / k * 1e3 for d in e["dts"]],
                "roofline_frac": (eb / (e["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if e["kern_ms"] else None,
                "occupied_fraction": e["occ"]}
        except Exception as ex:  # e.g. not enough memory on a shared box
            out["extra"] = {"error": str(ex)}

    if rank == 0 and world == 1 and not args.no_cull and not args.no_ablation:
        # Ablation in the same run: the same kernel with the rectangle tests off, i.e.
        # every voxel projected exactly in every view until it is carved (wave ballot
        # early-out only).  This is the closest thing to the per-view streaming
        # formulation the algorithmic byte count describes.
        try:
            k = max(3, args.steps // 4)
            b = run_config(args.grid, args.views, k, 1, "none", no_cull=True)
            bb = b["nvox"] * b["V"] + b["V"] * b["sc"].W * b["sc"].H
            out["ablation_no_cull"] = {
                "value": b["nvox"] * b["V"] / (b["dt"] / k) / 1e6, "unit": "Mvoxel-views/s",
                "carve_kernel_ms": b["kern_ms"],
                "roofline": {"bound": "hbm",
                             "achieved": (bb / (b["kern_ms"] * 1e-3) / 1e9) if b["kern_ms"] else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": (bb / (b["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if b["kern_ms"] else None}}
            if args.extra_grid and args.extra_grid != args.grid:
                # ... and on the north star's own configuration (1024^3 x 36 on one GPU: the target
                # is 40 % of the HBM-read roofline on the algorithmic bytes, i.e. <= 12.1 ms):
                # the brute-force kernel, every voxel in every view
                k2 = 3
                b2 = run_config(args.extra_grid, args.views, k2, 1, "none", no_cull=True)
                bb2 = b2["nvox"] * b2["V"] + b2["V"] * b2["sc"].W * b2["sc"].H
                out["ablation_no_cull"]["extra_grid"] = {
                    "workload": f"{args.extra_grid}^3 x {args.views} views, every voxel in every view",
                    "value": b2["nvox"] * b2["V"] / (b2["dt"] / k2) / 1e6, "unit": "Mvoxel-views/s",
                    "carve_kernel_ms": b2["kern_ms"], "ms_per_step": b2["dt"] / k2 * 1e3,
                    "roofline": {"bound": "hbm",
                                 "achieved": (bb2 / (b2["kern_ms"] * 1e-3) / 1e9) if b2["kern_ms"] else None,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (bb2 / (b2["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if b2["kern_ms"] else None}}
        except Exception as ex:
            out["ablation_no_cull"] = {"error": str(ex)}

    if rank == 0 and world == 1 and not args.no_e2e:
        # SURVEY 8d's second figure: the step with its inputs coming from and its result going to
        # the HOST -- page-locked u8 masks -> device (arvx_set_views), carve, the carved model back
        # in page-locked memory: as two compressed packets (arvx_state_download_packets; ~1/10 of the
        # bytes) and, beside it, as two plain bit planes (N / 4 bytes, arvx_state_download_planes)
        def e2e_entry(N):
            sc_ = synthetic.sphere_scene(N, args.views)
            masks = torch.from_numpy(sc_.masks).pin_memory()
            words = ((N + 31) // 32) * N * N
            occ = torch.empty(words, dtype=torch.int32).pin_memory()
            seen = torch.empty(words, dtype=torch.int32).pin_memory()
            res = {}
            with capi.Context(N, N, N, sc_.voxel_size, device=dev.index) as c:
                mnp = masks.numpy()
                n64, H = c.packet_geometry()
                # the packets' host buffers, sized by one untimed hand-off (a caller keeps them
                # from frame to frame: arvx::Model does)
                c.set_views(sc_.M, mnp)
                c.carve(0)
                _, _, need_o, need_s = c.download_packets()
                pk_o = torch.empty(H + need_o + need_o // 8, dtype=torch.int64).pin_memory()
                pk_s = torch.empty(H + need_s + need_s // 8 + 16, dtype=torch.int64).pin_memory()
                for form in ("packets", "planes"):
                    ms = []
                    for _ in range(8):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        c.reset()
                        c.set_views(sc_.M, mnp)
                        t1 = time.perf_counter()
                        c.carve(0)
                        c.synchronize()
                        t2 = time.perf_counter()
                        if form == "packets":
                            po, ps, no, ns = c.download_packets(pk_o.numpy().view(np.uint64),
                                                                pk_s.numpy().view(np.uint64))
                        else:
                            c.download_planes(occ.numpy().view(np.uint32), seen.numpy().view(np.uint32))
                        t3 = time.perf_counter()
                        ms.append(((t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
                    a = np.array(ms[3:])  # (the first rounds pay the first uses: staging buffers, pools)
                    res[form] = a
                # the two forms hold the same model (untimed): every mixed word at its place, the
                # bitmaps against the planes' all-zero / all-one words
                o64, s64 = occ.numpy().view(np.uint64), seen.numpy().view(np.uint64)
                same = True
                for pk, pl, need in ((po, o64, no), (ps, s64, ns)):
                    nb = (n64 + 63) // 64
                    ones = np.unpackbits(pk[1:1 + nb].view(np.uint8), bitorder="little")[:n64].astype(bool)
                    mixed = np.unpackbits(pk[1 + nb:1 + 2 * nb].view(np.uint8), bitorder="little")[:n64].astype(bool)
                    full = np.where(ones, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0))
                    full[mixed] = pk[H:H + need]
                    same = same and int(pk[0]) == need == int(mixed.sum()) and bool(np.array_equal(full, pl))
            a = res["packets"]
            tot = float(np.median(a[:, 0]))
            pl = res["planes"]
            return {"grid": [N, N, N], "views": args.views, "ms_total": tot,
                    "ms_masks_h2d_and_views": float(np.median(a[:, 1])),
                    "ms_carve": float(np.median(a[:, 2])),
                    "ms_model_d2h": float(np.median(a[:, 3])),
                    "value": N ** 3 * args.views / (tot * 1e-3) / 1e6, "unit": "Mvoxel-views/s",
                    "bytes_h2d": int(sc_.masks.nbytes),
                    "bytes_d2h": int(8 * (2 * H + need_o + need_s)),
                    "packets_equal_planes": same,
                    "planes_form": {"ms_total": float(np.median(pl[:, 0])),
                                    "ms_planes_d2h": float(np.median(pl[:, 3])),
                                    "bytes_d2h": int(2 * words * 4)}}
        try:
            out["e2e"] = {"what": "page-locked host masks -> arvx_set_views -> arvx_carve -> the carved "
                                  "model (occupancy + seen) into page-locked memory as compressed packets "
                                  "(arvx_state_download_packets: what arvx::Model reads; `planes_form`: "
                                  "the same as two plain bit planes, arvx_state_download_planes); median "
                                  "of 5 rounds after 3 warm ones; never part of `value`",
                          "runs": [e2e_entry(args.grid)] +
                                  ([e2e_entry(args.extra_grid)] if args.extra_grid and
                                   args.extra_grid != args.grid else [])}
        except Exception as ex:  # noqa: BLE001
            out["e2e"] = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"], oracle_plane = cpu_baseline(r["sc"], r["X"], r["Y"], r["Z"])
        # the state the timed steps left on the GPU against the oracle's, voxel for voxel
        nz = oracle_plane.shape[0]
        same = bool(np.array_equal(r["state"][:nz], oracle_plane))
        out["parity_vs_oracle"] = same
        out["parity_planes"] = f"z=0..{nz - 1} of {r['Z']}"
        if not same:
            print("[bench] GPU state differs from the oracle in "
                  f"{int((r['state'][:nz] != oracle_plane).sum())} voxels", file=sys.stderr)
    elif rank == 0:
        out["cpu_baseline"] = None
        out["parity_vs_oracle"] = None

    if rank == 0 and world == 1 and not args.no_workloads:
        try:
            out["workloads"] = run_workloads(capi, synthetic, dev, check=not args.no_cpu)
        except Exception as ex:  # noqa: BLE001
            out["workloads"] = [{"error": f"{type(ex).__name__}: {ex}"}]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
