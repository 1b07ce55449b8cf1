"""Headroom probe: the C4 scene with 8x / 32x the triangles (8 M, 32 M): scene creation, BVH size, one 1080p launch, and the ray service's two
kernels against each other and (8 M only) against the checker's tree.  usage: python tools/probe_large_scene.py [million triangles ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pearray_amd import backend, scene
for m in [int(a) for a in sys.argv[1:]] or [8]:
    t0 = time.time(); sc = scene.cornell_soup(1920, 1080, spp=4, n_triangles=m * 1_000_000); t1 = time.time()
    g = backend.RenderContext(sc); t2 = time.time()
    g.render(2); g.waitForFinish(); t3 = time.time()
    g.render(4); g.waitForFinish(); t4 = time.time()
    st = g.statistics()
    rng = np.random.default_rng(5); n = 200_000
    org = (rng.random((n, 3)) * [1.9, 1.9, 1.85] + [-0.95, -0.95, 0.05]).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    os.environ["PRGPU_TRACE_SPLIT"] = "1"; a = g.traceRays(org, d, 1e-4, np.inf)
    os.environ["PRGPU_TRACE_SPLIT"] = "0"; b = g.traceRays(org, d, 1e-4, np.inf)
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    print("%d M triangles: host arrays %.1f s, scene create %.2f s, 4 iterations %.1f ms each (%.1f Msamples/s), finite frame %s, pixel_samples %d; ray service kernels agree: %s"
          % (m, t1 - t0, t2 - t1, (t4 - t3) / 4 * 1e3, 1920 * 1080 * 4 / (t4 - t3) / 1e6, bool(np.isfinite(g.output()[0]).all()), st["pixel_samples"], same), flush=True)
    if m <= 8:
        import oracle_binding as ob
        t5 = time.time(); o = ob.OracleScene(sc); t6 = time.time()
        k = 20000
        c = o.trace_closest(org[:k], d[:k], 1e-4, np.inf)
        print("   checker tree built in %.1f s; %d rays: ids and distances equal: %s" % (t6 - t5, k, all(np.array_equal(x[:k], y) for x, y in zip(a, c))), flush=True)
    g.close()
