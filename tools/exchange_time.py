"""Time the compressed occupancy packet on one Z slab (GPU box): carve, pack, compress,
and the expand of `world` packets; prints the share of mixed words and the packet size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ar_voxel_project_amd import capi, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = syn.sphere_scene(N, 36)
stream = torch.cuda.Stream()
n = N * N * (N // world) // 64
for rank in (0, world // 2):
    z0, z1 = rank * N // world, (rank + 1) * N // world
    with capi.Context(N, N, N, sc.voxel_size, z_range=(z0, z1)) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_views(sc.M, sc.masks)
        with torch.cuda.stream(stream):
            words = torch.zeros(n, dtype=torch.int64, device="cuda")
            full = torch.zeros(n * world, dtype=torch.int64, device="cuda")
            flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            ctx.carve()
            ctx.pack_occupancy(words.data_ptr())
            pk = torch.zeros(capi.occupancy_packet_words(n, n), dtype=torch.int64, device="cuda")
            ctx.occupancy_compress(words.data_ptr(), n, pk.data_ptr(), n)
            stream.synchronize()
            need = int(pk[0])
            cap = need + need // 4 + 16
            S = capi.occupancy_packet_words(n, cap)
            pks = torch.zeros(world * S, dtype=torch.int64, device="cuda")
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            best = [1e9, 1e9]
            for _ in range(6):
                ev[0].record(stream)
                for q in range(world):
                    ctx.occupancy_compress(words.data_ptr(), n, pks[q * S:].data_ptr(), cap)
                ev[1].record(stream)
                ctx.occupancy_expand(pks.data_ptr(), world, 0, n, cap, full.data_ptr(), flag.data_ptr())
                ev[2].record(stream)
                stream.synchronize()
                best[0] = min(best[0], ev[0].elapsed_time(ev[1]) / world)
                best[1] = min(best[1], ev[1].elapsed_time(ev[2]))
            ok = all(torch.equal(full[q * n:(q + 1) * n], words) for q in range(1, world))
        print(f"N={N} slab {z0}:{z1}  words {n}  mixed {need} ({100.0 * need / n:.1f} %)  "
              f"packet {S * 8 / 1e6:.2f} MB vs plain {n * 8 / 1e6:.2f} MB  "
              f"compress {best[0]:.3f} ms  expand x{world} {best[1]:.3f} ms  "
              f"flag {int(flag)}  round-trip {ok}", flush=True)
