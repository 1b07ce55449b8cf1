"""How much could a better BVH give on the C4 scene?  The same rays through (a) the device LBVH (4-wide quantised nodes, leaves of
<= 3 triangles; record counters of the ray-service kernel) and (b) the checker's binned-SAH BVH2 (leaves of <= 4 triangles; its
node / triangle counters).  Rays: the paths' own mix -- camera rays, then rays from surface points into the cosine hemisphere.
usage: python tools/gpu_bvh_quality.py [n_triangles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
from oracle_binding import OracleScene
from pearray_amd import backend, scene

ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = scene.cornell_soup(1920, 1080, spp=4, n_triangles=ntri)
g = backend.RenderContext(sc)
o = OracleScene(sc)
rng = np.random.default_rng(7)
N = 200_000
# bounce-like rays: origins on the soup's triangles (hit points of random rays), cosine-free uniform directions
org0 = np.stack([rng.uniform(-0.9, 0.9, N), rng.uniform(-0.9, 0.9, N), rng.uniform(0.1, 1.8, N)], 1).astype(np.float32)
d0 = rng.normal(size=(N, 3)); d0 = (d0 / np.linalg.norm(d0, axis=1, keepdims=True)).astype(np.float32)
ent, prim, u, v, t = g.traceRays(org0, d0, 1e-4, np.inf)
hit = ent != 0xFFFFFFFF
org = (org0 + d0 * t[:, None])[hit] - 1e-3 * d0[hit]
d1 = rng.normal(size=(len(org), 3)); d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
org = org.astype(np.float32)


def device_counts(orgs, dirs):
    a = g.traceCounters()
    res = g.traceRays(orgs, dirs, 1e-4, np.inf)
    b = g.traceCounters()
    n = len(orgs)
    return res, (b["nodes_closest"] - a["nodes_closest"]) / n, (b["leaves_closest"] - a["leaves_closest"]) / n


def oracle_counts(orgs, dirs):
    n0, t0 = C.c_uint64(), C.c_uint64()
    o.lib.orc_trace_counters(o.h, C.byref(n0), C.byref(t0))
    res = o.trace_closest(orgs, dirs, 1e-4, np.inf)
    n1, t1 = C.c_uint64(), C.c_uint64()
    o.lib.orc_trace_counters(o.h, C.byref(n1), C.byref(t1))
    return res, (n1.value - n0.value) / len(orgs), (t1.value - t0.value) / len(orgs)


for name, oo, dd in (("uniform interior rays", org0, d0), ("bounce rays from surface points", org, d1)):
    rd, dn, dl = device_counts(oo, dd)
    ro, on, ot = oracle_counts(oo, dd)
    same = all(np.array_equal(x, y) for x, y in zip(rd[:2], ro[:2]))
    print("%-34s %d rays, hit ids equal=%s | device LBVH4: %.1f inner (64 B) + %.1f leaf (128 B) records = %.0f B/ray | checker SAH BVH2: %.1f nodes visited, %.1f triangles tested "
          "(as 4-wide nodes ~%.1f inner; leaves of <=4 ~%.1f leaf visits) " % (name, len(oo), same, dn, dl, 64 * dn + 128 * dl, on, ot, on / 2.2, ot / 2.6), flush=True)
