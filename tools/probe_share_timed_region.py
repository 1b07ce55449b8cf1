"""The driver's timed region for one rank of an 8-rank job, on one GPU: 5 warm-up iterations, then `render(20)` + `waitForFinish()` as bench.py
does -- with and without the pixel-order tuning that prgpu_sync runs at a synchronisation point (tune_pixel_order).
usage: python tools/probe_share_timed_region.py [world] [steps] [warmup]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import backend, scene, tiling
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
W, H = 1920, 1080
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
for env in ({}, {"PRGPU_PP_TUNE_ORDER": "0"}, {}, {"PRGPU_PP_TUNE_ORDER": "0"}):
    os.environ.pop("PRGPU_PP_TUNE_ORDER", None)
    os.environ.update(env)
    ctx = backend.RenderContext(sc)
    if world > 1:
        ctx.setTiles(tiling.tiles_for_rank(W, H, min(2, world - 1), world, tile=64 if world <= 2 else 16))
    ctx.render(warm); t = time.time(); ctx.waitForFinish(); t_sync1 = time.time() - t
    t = time.time(); ctx.render(steps); t_issue = time.time() - t; ctx.waitForFinish(); dt = time.time() - t
    print("%-28s rank %d of %d: warm-up sync %.2f ms; timed region %.2f ms = %.3f ms per step (render call returned after %.2f ms)"
          % (env or "defaults", min(2, world - 1), world, t_sync1 * 1e3, dt * 1e3, dt / steps * 1e3, t_issue * 1e3), flush=True)
    ctx.close()
