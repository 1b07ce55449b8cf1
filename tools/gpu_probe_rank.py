"""Probe: one rank's share of the C4 frame at world sizes 1/2/4/8 (strong-scaling proxy on a single GPU)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling
W, H, iters = 1920, 1080, 16
for world in (1, 2, 4, 8):
    sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
    ctx = backend.RenderContext(sc)
    tiles = tiling.tiles_for_rank(W, H, 0, world) if world > 1 else []
    ctx.setTiles(tiles)
    ctx.render(2); ctx.waitForFinish()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
    n = ctx.statistics()["pixel_samples"] / (iters + 2)
    print("world %d: rank-0 share %.0f px, %.2f ms/iter -> %.1f Msamples/s per GPU, x%d = %.1f (ideal-scaling proxy %.2fx)" % (
        world, n, dt / iters * 1e3, n * iters / dt / 1e6, world, world * n * iters / dt / 1e6, 0))
    ctx.close()
