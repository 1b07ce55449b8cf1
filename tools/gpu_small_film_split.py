"""Time split of the instrumented throughput kernel on a small film of the 1M-triangle scene (a launch that is all chain latency).
usage: PRGPU_DEBUG_COUNTERS=1 python tools/gpu_small_film_split.py [width]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene
w = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = backend.RenderContext(scene.cornell_soup(w, w, spp=256, n_triangles=1_000_000))
ctx.render(8); ctx.waitForFinish()
t = time.time(); ctx.render(32); ctx.waitForFinish(); dt = (time.time() - t) / 32 * 1e3
a = ctx.traceCounters(); s0 = ctx.statistics()
ctx.setInstrumentation(True); ctx.render(32); ctx.waitForFinish(); ctx.setInstrumentation(False)
b = ctx.traceCounters(); s1 = ctx.statistics()
d = {k: b[k] - a[k] for k in b if isinstance(b[k], int)}
n = s1["pixel_samples"] - s0["pixel_samples"]
print("%dx%d: %.3f ms/iteration; mean depth %.2f; wave steps per iteration %.1f k, shading passes %.1f k (fill %.3f)"
      % (w, w, dt, (s1["camera_depth"] - s0["camera_depth"]) / n, d["wave_steps_closest"] / 32 / 1e3, d["shade_batches"] / 32 / 1e3, d["shade_lanes"] / max(64 * d["shade_batches"], 1)))
