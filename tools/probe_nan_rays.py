"""How the ray service treats rays with a NaN / infinite component (a degenerate normal upstream can make one): hits and time per ray."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene
sc = scene.cornell_soup(64, 64, spp=1, n_triangles=1_000_000)
g = backend.RenderContext(sc)
rng = np.random.default_rng(1)
def run(name, org, d):
    g.traceRays(org[:1], d[:1], 1e-4, np.inf)
    t = time.time(); r = g.traceRays(org, d, 1e-4, np.inf); dt = time.time() - t
    print("%-34s %6d rays: %8.2f ms, %d hits" % (name, len(org), dt * 1e3, int((r[0] != 0xFFFFFFFF).sum())), flush=True)
n = 64
org = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (n, 1))
good = rng.normal(size=(n, 3)).astype(np.float32); good /= np.linalg.norm(good, axis=1, keepdims=True)
run("ordinary rays", org, good)
for name, comp in (("NaN in d.x", (0,)), ("NaN in all of d", (0, 1, 2))):
    d = good.copy(); d[:, comp] = np.nan
    run(name, org, d)
d = good.copy(); d[:, 1] = np.inf
run("inf in d.y", org, d)
o2 = org.copy(); o2[:, 2] = np.nan
run("NaN in origin.z", o2, good)
