"""Latency of one path vertex when hardly anything else runs: a few pixels of a glass-heavy scene rendered for many iterations -- a pixel's
samples are a chain through its RNG stream, so the time per vertex of such a render is what bounds small tile shares.
usage: python tools/gpu_chain_latency.py [pixels per side] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for name, sc in (("glass Cornell box", scene.cornell_glassy(n, n, spp=1024)), ("Lambert Cornell box", scene.cornell_box(n, n, spp=1024)),
                 ("1M-triangle Cornell box", scene.cornell_soup(n, n, spp=1024, n_triangles=1_000_000))):
    ctx = backend.RenderContext(sc)
    ctx.render(8); ctx.waitForFinish()
    s0 = ctx.statistics()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
    s1 = ctx.statistics()
    verts = s1["camera_depth"] - s0["camera_depth"] + s1["background_hits"] - s0["background_hits"]
    samples = s1["pixel_samples"] - s0["pixel_samples"]
    rays = sum(s1[k] - s0[k] for k in ("primary_rays", "bounce_rays", "shadow_rays"))
    # the render lasts as long as the pixel with the most vertices; mean over pixels as a stand-in (tiny film: similar pixels)
    per_pixel = verts / (n * n)
    print("%-24s %dx%d pixels, %d iterations: %.1f ms, %.1f vertices per sample, %.1f us per vertex of a pixel's chain, %.1f us per iteration"
          % (name, n, n, iters, dt * 1e3, verts / max(samples, 1), dt * 1e6 / max(per_pixel, 1), dt * 1e6 / iters), flush=True)
    ctx.close()
