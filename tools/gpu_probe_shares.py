"""Strong-scaling proxy on ONE GPU: time rank 0's tile share of the C4 frame for world sizes 1, 2, 4, 8 (default pipeline).
usage: python tools/gpu_probe_shares.py [iterations ...]   (run on the GPU box under `timeout`)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
for iters in [int(a) for a in sys.argv[1:]] or [16, 96]:
    base = None
    for world in (1, 2, 4, 8):
        ctx = backend.RenderContext(sc)
        if world > 1:
            ctx.setTiles(tiling.tiles_for_rank(W, H, 0, world))
        ctx.render(8); ctx.waitForFinish()
        t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = (time.time() - t) / iters * 1e3
        ctx.close()
        base = base or dt
        print("iters %3d  share 1/%d  %.2f ms/iteration  (%.2fx)" % (iters, world, dt, base / dt), flush=True)
