"""Instrumented persistent-kernel counters for a tile share of the C4 (or C5) frame: records per ray, wave steps, lane utilisation, time split.
usage: python tools/gpu_counters.py [world] [iterations] [scene: c4|c5]   (GPU box; env knobs apply)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 16
which = sys.argv[3] if len(sys.argv) > 3 else "c4"
if which == "c5":
    sc = scene.ArrayScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = W, H
else:
    sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
if world > 1:
    ctx.setTiles(tiling.tiles_for_rank(W, H, 0, world, tile=64 if world <= 2 else 16))
ctx.render(8); ctx.waitForFinish()
t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = (time.time() - t) / iters * 1e3
a = ctx.traceCounters()
ctx.setInstrumentation(True); ctx.render(iters); ctx.waitForFinish(); ctx.setInstrumentation(False)
b = ctx.traceCounters()
d = {k: b[k] - a[k] for k in b if isinstance(b[k], int)}
rc, ra = max(d["rays_closest"], 1), max(d["rays_any"], 1)
steps = d["wave_steps_closest"]   # (the persistent kernel counts every wave step here, whichever kind of ray its lanes hold)
recs = d["nodes_closest"] + d["leaves_closest"] + d["nodes_any"] + d["leaves_any"]
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("PRGPU_"))
info = ctx.pipelineInfo()
print("[%s] %s share 1/%d: %.3f ms/iteration (plain kernel); BVH %d-wide (estimates: 4-wide %.2f, 6-wide %.2f inner records per ray through the scene's box)"
      % (tag, which, world, dt, info["bvh_width"], info["bvh_cost_4_wide"], info["bvh_cost_6_wide"]))
print("  per iteration: closest rays %.2f M, occlusion rays %.2f M; inner/leaf records per closest ray %.2f / %.2f, per occlusion ray %.2f / %.2f"
      % (rc / iters / 1e6, ra / iters / 1e6, d["nodes_closest"] / rc, d["leaves_closest"] / rc, d["nodes_any"] / ra, d["leaves_any"] / ra))
print("  wave steps per iteration %.2f M; records per step %.1f (lane utilisation %.3f)"
      % (steps / iters / 1e6, recs / max(steps, 1), recs / max(64 * steps, 1)))
print("  shading passes per iteration %.1f k, fill %.3f; wave time: shading %.1f %%, idle %.1f %%"
      % (d["shade_batches"] / iters / 1e3, d["shade_lanes"] / max(64 * d["shade_batches"], 1), 100.0 * d["shade_ticks"] / max(d["total_ticks"], 1), 100.0 * d["idle_ticks"] / max(d["total_ticks"], 1)))
