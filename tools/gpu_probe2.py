"""Probe: per-launch floor of the traversal kernels (few rays) -- tail ray or fixed launch cost?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene
sc = scene.cornell_soup(256, 144, spp=4, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
rng = np.random.default_rng(3)
for n in (64, 1024, 16384, 262144, 1048576):
    org = (rng.random((n, 3)) * [1.8, 1.8, 1.7] + [-0.9, -0.9, 0.1]).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ctx.traceRays(org, d, 1e-4, np.inf)
    ctx.setTiming(True)
    for _ in range(5):
        ctx.traceRays(org, d, 1e-4, np.inf)
    ms, k = ctx.kernelTime("trace_closest")
    ctx.setTiming(False)
    print("n=%8d  closest service kernel: %.3f ms avg over %d (cumulative counters)" % (n, ms / k, k))
