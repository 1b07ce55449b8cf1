"""Ad-hoc GPU probe: time the wavefront on the C4 scene (1M-triangle Cornell soup) for a few iterations."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene

W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1920, int(sys.argv[2]) if len(sys.argv) > 2 else 1080
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ntri = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
t = time.time(); sc = scene.cornell_soup(W, H, spp=1024, n_triangles=ntri); print("scene assembly %.2fs" % (time.time() - t))
t = time.time(); ctx = backend.RenderContext(sc); print("scene_create (upload+LBVH+tables) %.2fs" % (time.time() - t))
ctx.render(1); ctx.waitForFinish()
ctx.setTiming(True)
t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
st = ctx.statistics()
print("iters %d: %.3fs -> %.2f Msamples/s" % (iters, dt, W * H * iters / dt / 1e6))
for fam in ("raygen", "trace_closest", "shade", "trace_any", "resolve"):
    ms, n = ctx.kernelTime(fam)
    print("  %-14s %8.2f ms over %5d launches" % (fam, ms, n))
print(st)
rays = st["primary_rays"] + st["bounce_rays"] + st["shadow_rays"]
print("rays/sample %.2f  mean depth %.2f" % (rays / st["pixel_samples"], st["camera_depth"] / st["pixel_samples"]))
ctx.setTiming(False); ctx.setInstrumentation(True)
ctx.render(1); ctx.waitForFinish()
print(ctx.traceCounters())
xyz, smp, fb = ctx.output()
print("mean xyz", xyz.reshape(-1, 3).mean(0), "feedback any", int((fb != 0).sum()))
tc = ctx.traceCounters()
rc = (tc["nodes_closest"] + tc["leaves_closest"]); ra = (tc["nodes_any"] + tc["leaves_any"])
n_c = (st["primary_rays"] + st["bounce_rays"]) / (iters + 1); n_a = st["shadow_rays"] / (iters + 1)
print("closest: %.1f inner + %.1f leaf records/ray, lane utilisation per step %.3f ; any: %.1f + %.1f records/ray, utilisation %.3f" % (
    tc["nodes_closest"] / n_c, tc["leaves_closest"] / n_c, rc / max(1, 64 * tc["wave_steps_closest"]),
    tc["nodes_any"] / n_a, tc["leaves_any"] / n_a, ra / max(1, 64 * tc["wave_steps_any"])))
