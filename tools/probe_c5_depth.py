"""Why small tile shares of C5 do not scale: a pixel's samples are a chain (its RNG stream), and on the glass objects some pixels run to the
depth limit in every sample.  (1) rank 2's 1/8 share with the limit at 64 / 16 / 8 (diagnosis only: another limit is another image);
(2) the deepest pixels of that share rendered ALONE (a tile of one pixel each): the per-vertex latency of a lone path in this scene.
usage: python tools/probe_c5_depth.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pearray_amd import backend, scene, tiling
W, H = 1920, 1080


def load(depth=64):
    sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = W, H
    sc.desc.settings.max_ray_depth = depth
    return sc


deep = None
for depth in (64, 16, 8):
    ctx = backend.RenderContext(load(depth))
    ctx.setTiles(tiling.tiles_for_rank(W, H, 2, 8, tile=16))
    ctx.render(8); ctx.waitForFinish()
    t = time.time(); ctx.render(32); ctx.waitForFinish(); dt = (time.time() - t) / 32 * 1e3
    c = np.asarray(ctx.pathCost()).reshape(H, W).astype(np.float64) / 40.0
    if depth == 64:
        ys, xs = np.unravel_index(np.argsort(c, axis=None)[-4:], c.shape)
        deep = list(zip(xs.tolist(), ys.tolist(), c[ys, xs].tolist()))
    v = c[c > 0]
    print("C5 rank 2 of 8, max_ray_depth %d: %.3f ms per iteration; vertices per sample and pixel: mean %.2f p99 %.1f p99.9 %.1f max %.1f"
          % (depth, dt, v.mean(), np.percentile(v, 99), np.percentile(v, 99.9), v.max()), flush=True)
    ctx.close()
for n_px in (1, 4):
    ctx = backend.RenderContext(load())
    ctx.setTiles([(deep[-1 - k][0], deep[-1 - k][1], deep[-1 - k][0] + 1, deep[-1 - k][1] + 1) for k in range(n_px)])
    ctx.render(8); ctx.waitForFinish()
    s0 = ctx.statistics()
    t = time.time(); ctx.render(64); ctx.waitForFinish(); dt = time.time() - t
    s1 = ctx.statistics()
    verts = s1["camera_depth"] - s0["camera_depth"] + s1["background_hits"] - s0["background_hits"]
    print("the %d deepest pixel(s) of that share alone, 64 iterations: %.2f ms per iteration, %.1f vertices per sample, %.1f us per vertex of a pixel's chain"
          % (n_px, dt / 64 * 1e3, verts / (64.0 * n_px), dt * 1e6 / max(verts / n_px, 1)), flush=True)
    ctx.close()
