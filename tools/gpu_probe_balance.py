"""Load balance of the tile dealing on ONE GPU: ms per iteration of EVERY rank's share for a world size and several tile sizes.
usage: python tools/gpu_probe_balance.py [world] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
for tile in (64, 32, 16, 8):
    ts = []
    for rank in range(world):
        ctx = backend.RenderContext(sc)
        tl = tiling.tiles_for_rank(W, H, rank, world, tile=tile)
        ctx.setTiles(tl)
        ctx.render(4); ctx.waitForFinish()
        t = time.time(); ctx.render(iters); ctx.waitForFinish(); ts.append((time.time() - t) / iters * 1e3)
        ctx.close()
    print("tile %2d world %d: " % (tile, world) + " ".join("%.2f" % t for t in ts) + "  | max %.2f mean %.2f" % (max(ts), sum(ts) / len(ts)), flush=True)
