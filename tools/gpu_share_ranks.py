"""Every rank's 1/world share of the C4 frame, one after the other on one GPU: round-robin Z-order deal at several tile sizes, and a
cost-balanced deal (a pilot pass measures the path vertices per tile, tiles go to ranks longest-processing-time-first).
usage: python tools/gpu_share_ranks.py [world] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 96
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)


def run(tiles, want_cost=False):
    ctx = backend.RenderContext(sc)
    ctx.setTiles(tiles)
    ctx.render(8); ctx.waitForFinish()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = (time.time() - t) / iters * 1e3
    cost = ctx.pathCost() if want_cost else None
    ctx.close()
    return dt, cost


for tile in [int(a) for a in os.environ.get("TILES", "8,16,32").split(",")]:
    times, cost = [], np.zeros((H, W), dtype=np.uint64)
    for rank in range(world):
        dt, c = run(tiling.tiles_for_rank(W, H, rank, world, tile=tile), want_cost=True)
        times.append(dt); cost += c
    print("tile %2d round robin : max %.3f mean %.3f min %.3f ms/iteration | %s" % (tile, max(times), np.mean(times), min(times), " ".join("%.3f" % t for t in times)), flush=True)
    if not cost.any():
        print("        (no cost plane: the share is not all in flight)"); continue
    tiles = tiling.all_tiles(W, H, tile)
    tc = np.array([cost[y0:y1, x0:x1].sum() for x0, y0, x1, y1 in tiles], dtype=np.float64)
    for name, key in (("vertices", tc), ("vertices + pixels", tc / tc.sum() + np.array([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles]) / float(W * H))):
        load, deal = np.zeros(world), [[] for _ in range(world)]
        for k in np.argsort(-key):
            r = int(np.argmin(load)); load[r] += key[k]; deal[r].append(k)
        times = []
        for r in range(world):
            dt, _ = run([tiles[k] for k in sorted(deal[r])])
            times.append(dt)
        print("tile %2d LPT by %-18s: max %.3f mean %.3f min %.3f ms/iteration | load spread %.4f | %s" % (tile, name, max(times), np.mean(times), min(times), load.max() / load.mean(), " ".join("%.3f" % t for t in times)), flush=True)
