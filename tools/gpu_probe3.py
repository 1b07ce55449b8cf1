"""Probe: iteration throughput without per-kernel timing, for PRGPU_GROUPS sweeps."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene
W, H, iters = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
ctx.render(2); ctx.waitForFinish()
t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
print("groups=%s iters %d: %.3fs -> %.2f Msamples/s (%.2f ms/iter)" % (os.environ.get("PRGPU_GROUPS", "default"), iters, dt, W * H * iters / dt / 1e6, dt / iters * 1e3))
