"""Ray service (prgpu_trace_closest, the IArchive surface) on the C4 scene: Mrays/s of its kernel variants on the same incoherent rays, with
records per ray, wave steps and lane utilisation, and a check that every variant returns the same hits.
usage: python tools/gpu_ray_service.py [million rays]     variants: split traversal (default), classic (PRGPU_TRACE_SPLIT=0).
(PRGPU_TRACE_TWO_RAYS=1 selected round 4's two-rays-per-lane experiment: profiles/r04_two_rays_per_lane.{log,patch}; the kernel is not in the tree.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene

N = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 8_000_000
sc = scene.cornell_soup(1920, 1080, spp=4, n_triangles=1_000_000)
rng = np.random.default_rng(11)
org = np.stack([rng.uniform(-0.9, 0.9, N), rng.uniform(-0.9, 0.9, N), rng.uniform(0.1, 1.8, N)], 1).astype(np.float32)
d = rng.normal(size=(N, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
ref = None
for name, env in (("split traversal (default)", {}), ("classic, one ray per lane", {"PRGPU_TRACE_SPLIT": "0"})):
    for k in ("PRGPU_TRACE_SPLIT", "PRGPU_TRACE_TWO_RAYS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = backend.RenderContext(sc)
    g.setTiming(True)
    g.traceRays(org[:1 << 20], d[:1 << 20], 1e-4, np.inf)     # warm-up
    a, (ms0, n0) = g.traceCounters(), g.kernelTime("trace_closest")
    res = g.traceRays(org, d, 1e-4, np.inf)
    b, (ms1, n1) = g.traceCounters(), g.kernelTime("trace_closest")
    ms = ms1 - ms0
    recs = b["nodes_closest"] - a["nodes_closest"] + b["leaves_closest"] - a["leaves_closest"]
    steps = b["wave_steps_closest"] - a["wave_steps_closest"]
    same = "reference" if ref is None else str(all(np.array_equal(x, y) for x, y in zip(ref, res)))
    ref = ref or res
    print("%-28s %5.2f ms for %.1f M rays = %6.0f Mrays/s | %.1f inner + %.1f leaf records per ray, %.2f M wave steps, records per lane-step %.3f | same hits: %s"
          % (name, ms, N / 1e6, N / ms / 1e3, (b["nodes_closest"] - a["nodes_closest"]) / N, (b["leaves_closest"] - a["leaves_closest"]) / N, steps / 1e6,
             recs / max(64 * steps, 1), same), flush=True)
    g.close()
