import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling
W, H, iters, world = 1920, 1080, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
ctx.setTiles(tiling.tiles_for_rank(W, H, 0, world) if world > 1 else [])
ctx.render(2); ctx.waitForFinish()
ctx.setTiming(True)
t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
print("world %d: %.2f ms/iter" % (world, dt / iters * 1e3))
for fam in ("raygen", "trace_closest", "shade", "trace_any", "resolve"):
    ms, n = ctx.kernelTime(fam)
    print("  %-14s %8.3f ms/iter over %5.1f launches/iter (avg %.3f ms)" % (fam, ms / iters, n / iters, ms / max(n, 1)))
