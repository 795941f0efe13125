import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ar_voxel_project_amd import capi, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
sc = synthetic.sphere_scene(N, 36)
d_masks = torch.from_numpy(sc.masks).to(dev)
for mode in ("plain", "views-high", "plain", "views-high"):
    slots = 4
    lo = [torch.cuda.Stream(device=dev, priority=0) for _ in range(slots)]
    hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(slots)]
    ctxs = [capi.Context(N, N, N, sc.voxel_size) for _ in range(slots)]
    for c, st in zip(ctxs, lo):
        c.set_stream(st.cuda_stream)
    best = None
    for rep in range(4):
        K = 120
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            s = k % slots
            c = ctxs[s]
            c.reset()
            if mode == "plain":
                c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1)
                c.carve(0)
            else:
                hi[s].wait_stream(lo[s])      # the previous carve of this slot read the tables
                c.set_stream(hi[s].cuda_stream)
                c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1)
                lo[s].wait_stream(hi[s])
                c.set_stream(lo[s].cuda_stream)
                c.carve(0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        best = ms if best is None or ms < best else best
    for c in ctxs:
        c.close()
    print(f"N={N} {mode}: {best:.4f} ms per step", flush=True)
