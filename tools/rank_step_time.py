#!/usr/bin/env python3
"""What ONE rank of an N-GPU job does per step, timed on one GPU without the collective:
views + carve of the rank's stripes of the (N x 512^3)-voxel grid, pack, compress, expand of
`world` packets (its own, repeated).  python tools/rank_step_time.py [world=8] [views=36]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ar_voxel_project_amd import capi, sharding, synthetic as syn  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 36
X, Y, Z = sharding.grid_for(world, 512)
sc = syn.sphere_scene(max(X, Y, Z), V)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
for rank in (0, world // 2):
    ctx = capi.Context(X, Y, Z, sc.voxel_size, stripes=(world, rank))
    ctx.set_stream(stream.cuda_stream)
    d_masks = torch.from_numpy(sc.masks).to(dev)
    ctx.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
    n64 = X * Y * (Z // world) // 64
    wpg = X * Y * 8 // 64
    local = torch.zeros(n64, dtype=torch.int64, device=dev)
    full = torch.zeros(n64 * world, dtype=torch.int64, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.reset(); ctx.carve(); ctx.pack_occupancy(local.data_ptr())
    pk = torch.zeros(capi.occupancy_packet_words(n64, n64), dtype=torch.int64, device=dev)
    ctx.occupancy_compress(local.data_ptr(), n64, pk.data_ptr(), n64)
    torch.cuda.synchronize()
    need = int(pk[0]); cap = need + need // 4 + 16
    S = capi.occupancy_packet_words(n64, cap)
    pks = torch.zeros(world * S, dtype=torch.int64, device=dev)
    for q in range(world):
        ctx.occupancy_compress(local.data_ptr(), n64, pks[q * S:].data_ptr(), cap)
    names = ["views", "carve", "pack", "compress", "expand"]
    best = [1e9] * 5
    for _ in range(8):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ctx.reset()
        ev[0].record(stream)
        ctx.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
        ev[1].record(stream)
        ctx.carve()
        ev[2].record(stream)
        ctx.pack_occupancy(local.data_ptr())
        ev[3].record(stream)
        ctx.occupancy_compress(local.data_ptr(), n64, pks.data_ptr(), cap)
        ev[4].record(stream)
        ctx.occupancy_expand_striped(pks.data_ptr(), world, n64, cap, wpg, full.data_ptr(), flag.data_ptr())
        ev[5].record(stream)
        torch.cuda.synchronize()
        for k in range(5):
            best[k] = min(best[k], ev[k].elapsed_time(ev[k + 1]))
    print(f"world {world} rank {rank}: grid {X}x{Y}x{Z} x {V} views, packet {S * 8 / 1e6:.2f} MB of "
          f"{n64 * 8 / 1e6:.2f} MB | " + " | ".join(f"{n} {b * 1e3:.1f} us" for n, b in zip(names, best))
          + f" | sum {sum(best) * 1e3:.1f} us", flush=True)
    # round 4: the packet straight from the records (own words into the plane), the expansion
    # without the rank's own packet
    names2 = ["views", "carve", "pack+compress", "expand others"]
    best2 = [1e9] * 4
    for _ in range(8):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ctx.reset()
        ev[0].record(stream)
        ctx.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
        ev[1].record(stream)
        ctx.carve()
        ev[2].record(stream)
        ctx.occupancy_pack_compress(pks[rank * S:].data_ptr(), cap, full.data_ptr())
        ev[3].record(stream)
        ctx.occupancy_expand_striped_others(pks.data_ptr(), world, rank, n64, cap, wpg, full.data_ptr(),
                                            flag.data_ptr())
        ev[4].record(stream)
        torch.cuda.synchronize()
        for k in range(4):
            best2[k] = min(best2[k], ev[k].elapsed_time(ev[k + 1]))
    print(f"world {world} rank {rank}: fused hand-off | "
          + " | ".join(f"{n} {b * 1e3:.1f} us" for n, b in zip(names2, best2))
          + f" | hand-off {sum(best2[2:]) * 1e3:.1f} us (was {sum(best[2:]) * 1e3:.1f})", flush=True)
    FUSED = os.environ.get("ARVX_RANK_STEP_OLD") != "1"
    # the same work as a pipeline: hand-off of step k on a side stream beside step k + 1
    import time
    side = torch.cuda.Stream(device=dev)

    def run(overlap, steps=100):
        ctx.set_exchange_stream(side.cuda_stream if overlap else 0)
        packed = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.reset()
            ctx.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
            if overlap and packed is not None:
                stream.wait_event(packed)
            ctx.carve()
            if overlap:
                carved = torch.cuda.Event()
                carved.record(stream)
                side.wait_event(carved)
            if FUSED:
                ctx.occupancy_pack_compress(pks[rank * S:].data_ptr(), cap, full.data_ptr())
            else:
                ctx.pack_occupancy(local.data_ptr())
            if overlap:
                packed = torch.cuda.Event()
                packed.record(side)
            if FUSED:
                ctx.occupancy_expand_striped_others(pks.data_ptr(), world, rank, n64, cap, wpg,
                                                    full.data_ptr(), flag.data_ptr())
            else:
                ctx.occupancy_compress(local.data_ptr(), n64, pks.data_ptr(), cap)
                ctx.occupancy_expand_striped(pks.data_ptr(), world, n64, cap, wpg, full.data_ptr(),
                                             flag.data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    for _ in range(2):
        a, b = run(False), run(True)
    print(f"world {world} rank {rank}: per step, hand-off behind the carve {a:.1f} us, beside the next "
          f"step {b:.1f} us", flush=True)
    ctx.set_exchange_stream(0)
    ctx.close()

    # ... and with several jobs in flight (bench.py --jobs): (A) the hand-offs on two side streams
    # as bench.py had them at first, one per packed buffer, ordered against the slots' streams by
    # events; (B) every slot does its own hand-off in its own stream, no event at all
    def run_slots(variant, slots=4, steps=120):
        cs = [capi.Context(X, Y, Z, sc.voxel_size, stripes=(world, rank)) for _ in range(slots)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(slots)]
        for c, st in zip(cs, sts):
            c.set_stream(st.cuda_stream)
            c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
            c.reset()
            c.carve()
        nb = 2 if variant == "A" else slots
        locals_ = [torch.zeros(n64, dtype=torch.int64, device=dev) for _ in range(nb)]
        pkss = [torch.zeros(world * S, dtype=torch.int64, device=dev) for _ in range(nb)]
        fulls = [torch.zeros(n64 * world, dtype=torch.int64, device=dev) for _ in range(nb)]
        sides = [torch.cuda.Stream(device=dev) for _ in range(2)]
        packed = [None] * slots
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                s_, c, st = k % slots, cs[k % slots], sts[k % slots]
                c.reset()
                c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
                if variant == "A":
                    b = k % 2
                    if packed[s_] is not None:
                        st.wait_event(packed[s_])
                    c.carve()
                    carved = torch.cuda.Event()
                    carved.record(st)
                    c.set_exchange_stream(sides[b].cuda_stream)
                    sides[b].wait_event(carved)
                    c.pack_occupancy(locals_[b].data_ptr())
                    packed[s_] = torch.cuda.Event()
                    packed[s_].record(sides[b])
                else:
                    b = s_
                    c.carve()
                    if not FUSED:
                        c.pack_occupancy(locals_[b].data_ptr())
                if FUSED and variant != "A":
                    c.occupancy_pack_compress(pkss[b][rank * S:].data_ptr(), cap, fulls[b].data_ptr())
                    c.occupancy_expand_striped_others(pkss[b].data_ptr(), world, rank, n64, cap, wpg,
                                                      fulls[b].data_ptr(), flag.data_ptr())
                else:
                    c.occupancy_compress(locals_[b].data_ptr(), n64, pkss[b].data_ptr(), cap)
                    c.occupancy_expand_striped(pkss[b].data_ptr(), world, n64, cap, wpg, fulls[b].data_ptr(),
                                               flag.data_ptr())
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
            best = us if best is None or us < best else best
        for c in cs:
            c.set_exchange_stream(0)
            c.close()
        return best

    for _ in range(2):
        print(f"world {world} rank {rank}: 4 jobs in flight, per step: hand-offs on two side streams "
              f"{run_slots('A'):.1f} us, in the slot's own stream {run_slots('B'):.1f} us", flush=True)

    # ... and one job after the other on ONE compute stream with TWO sets of state records taken in
    # turn (two contexts on that stream): the carve of step k + 1 does not wait for the pack of step
    # k, which reads the other set; the hand-offs run on two side streams (bench.py, N > 1, --jobs 1)
    def run_two_sets(steps=120):
        cs = [capi.Context(X, Y, Z, sc.voxel_size, stripes=(world, rank)) for _ in range(2)]
        for c in cs:
            c.set_stream(stream.cuda_stream)
            c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
            c.reset()
            c.carve()
        pkss = [torch.zeros(world * S, dtype=torch.int64, device=dev) for _ in range(2)]
        fulls = [torch.zeros(n64 * world, dtype=torch.int64, device=dev) for _ in range(2)]
        sides = [torch.cuda.Stream(device=dev) for _ in range(2)]
        packed = [None, None]
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                b, c = k % 2, cs[k % 2]
                c.reset()
                c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
                if packed[b] is not None:
                    stream.wait_event(packed[b])  # (the pack of step k - 2: long done)
                c.carve()
                carved = torch.cuda.Event()
                carved.record(stream)
                c.set_exchange_stream(sides[b].cuda_stream)
                sides[b].wait_event(carved)
                c.occupancy_pack_compress(pkss[b][rank * S:].data_ptr(), cap, fulls[b].data_ptr())
                packed[b] = torch.cuda.Event()
                packed[b].record(sides[b])
                c.occupancy_expand_striped_others(pkss[b].data_ptr(), world, rank, n64, cap, wpg,
                                                  fulls[b].data_ptr(), flag.data_ptr())
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
            best = us if best is None or us < best else best
        for c in cs:
            c.set_exchange_stream(0)
            c.close()
        return best

    print(f"world {world} rank {rank}: one job after the other, two sets of records in turn, hand-offs on two "
          f"side streams: {run_two_sets():.1f} us per step", flush=True)
