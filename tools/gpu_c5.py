"""BASELINE config C5 on one GPU: examples/complex.prc (fixture tests/golden/scenes/complex_c5.npz; the Hosek-Wilkie sky table rebuilt from the
stored parameters) at 1920x1080 -- one full-resolution iteration checked against the CPU oracle, then timed, with the kernel's
time split (traversal / shading) from the instrumented variant.  usage: python tools/gpu_c5.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_binding import OracleScene
from pearray_amd import backend, scene

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H = (int(os.environ.get("C5_W", 1920)), int(os.environ.get("C5_H", 1080)))
sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))
sc.desc.settings.width, sc.desc.settings.height = W, H
t = time.time(); ctx = backend.RenderContext(sc); t_create = time.time() - t
if os.environ.get("C5_CHECK", "1") != "0":
    ctx.render(1); ctx.waitForFinish()
    xyz, smp, fb = ctx.output(); ge, gp = ctx.primaryHits()
    ora = OracleScene(sc)
    t = time.time(); ora.render(1, threads=os.cpu_count()); t_cpu = time.time() - t
    oxyz, osmp, ofb = ora.output(); oe, op = ora.primary_hits()
    print("C5 %dx%d one iteration vs oracle: hit ids equal=%s frame bit-exact=%s samples equal=%s feedback equal=%s stats equal=%s | oracle %.3f Msamples/s on %d threads"
          % (W, H, bool(np.array_equal(ge, oe) and np.array_equal(gp, op)), bool(np.array_equal(xyz, oxyz)), bool(np.array_equal(smp, osmp)),
             bool(np.array_equal(fb, ofb)), ctx.statistics() == ora.statistics(), W * H / t_cpu / 1e6, os.cpu_count()), flush=True)
else:
    ctx.render(1); ctx.waitForFinish()
s0 = ctx.statistics()
t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
s1 = ctx.statistics()
n = s1["pixel_samples"] - s0["pixel_samples"]
rays = sum(s1[k] - s0[k] for k in ("primary_rays", "bounce_rays", "shadow_rays"))
print("C5 %dx%d: %.1f Msamples/s, %.0f Mrays/s, %.2f ms/iteration over %d iterations, mean depth %.2f, scene create %.2f s"
      % (W, H, n / dt / 1e6, rays / dt / 1e6, dt / iters * 1e3, iters, (s1["camera_depth"] - s0["camera_depth"]) / max(n, 1), t_create), flush=True)
tc0 = ctx.traceCounters(); ctx.setInstrumentation(True); ctx.render(4); ctx.waitForFinish(); ctx.setInstrumentation(False); tc1 = ctx.traceCounters()
d = {k: tc1[k] - tc0[k] for k in tc1 if isinstance(tc1[k], int)}
rec = d["nodes_closest"] + d["leaves_closest"] + d["nodes_any"] + d["leaves_any"]
print("   instrumented: shading %.1f %% of wave time, idle %.1f %%, lane utilisation %.3f, shade pass fill %.3f, %.1f inner + %.1f leaf records per closest ray"
      % (100.0 * d["shade_ticks"] / max(d["total_ticks"], 1), 100.0 * d["idle_ticks"] / max(d["total_ticks"], 1),
         rec / max(64 * (d["wave_steps_closest"] + d["wave_steps_any"]), 1), d["shade_lanes"] / max(64 * d["shade_batches"], 1),
         d["nodes_closest"] / max(d["rays_closest"], 1), d["leaves_closest"] / max(d["rays_closest"], 1)), flush=True)
