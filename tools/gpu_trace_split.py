"""The ray service's plain closest-hit kernel (PRGPU_TRACE_SPLIT=0) against its split traversal (the default: leaf tests handed to
whole waves through an LDS task queue) on incoherent rays inside the 1 M-triangle C4 scene -- identical results, kernel time side by side.
usage: python tools/gpu_trace_split.py [million rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 8_000_000
sc = scene.cornell_soup(64, 64, spp=1, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
rng = np.random.default_rng(7)
org = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(-0.95, 0.95, n), rng.uniform(0.05, 1.90, n)], axis=1).astype(np.float32)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = d.astype(np.float32)
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["PRGPU_TRACE_SPLIT"] = mode
    ctx.setTiming(True)
    t0 = ctx.kernelTime("trace_closest")[0]
    tc0 = ctx.traceCounters()
    out = ctx.traceRays(org, d, 1e-4, np.inf)
    ctx.waitForFinish()
    ms = ctx.kernelTime("trace_closest")[0] - t0
    tc1 = ctx.traceCounters()
    dn, dl, dw = (tc1[k] - tc0[k] for k in ("nodes_closest", "leaves_closest", "wave_steps_closest"))
    print("split=%s: %.2f ms for %d rays = %.0f Mrays/s | %.1f inner + %.1f leaf records per ray, %d wave steps" % (mode, ms, n, n / ms / 1e3, dn / n, dl / n, dw), flush=True)
    if mode in res:
        continue
    res[mode] = out
a, b = res["0"], res["1"]
names = ("entity", "prim", "u", "v", "t")
for k in range(5):
    same = np.array_equal(a[k], b[k])
    print("%s identical: %s" % (names[k], same) + ("" if same else "  (%d of %d differ)" % ((a[k] != b[k]).sum(), n)))
print("hit fraction %.3f" % (a[0] != 0xFFFFFFFF).mean())
