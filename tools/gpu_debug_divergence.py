"""Find the first pixel/iteration where GPU and oracle diverge and replay that pixel's rays on the GPU ray service."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_binding as ob
from pearray_amd import backend, scene, _cabi as abi

name = sys.argv[1] if len(sys.argv) > 1 else "eval"
sc = {"eval": lambda: scene.cbox_eval(64, 64, spp=8), "soup": lambda: scene.cornell_soup(256, 144, spp=2, n_triangles=1_000_000)}[name]()
g, o = backend.RenderContext(sc), ob.OracleScene(sc)
prev_g = prev_o = None
for it in range(sc.spp):
    g.render(1); g.waitForFinish(); o.render(1, threads=8)
    gx, ox = g.output()[0], o.output()[0]
    bad = np.argwhere((gx != ox).any(-1))
    print("iter", it, "differing pixels", len(bad), "stats equal", g.statistics() == o.statistics())
    if len(bad):
        y, x = bad[0]
        pix = int(y) * sc.width + int(x)
        print("first bad pixel", x, y, gx[y, x], ox[y, x])
        o2 = ob.OracleScene(sc)
        o2.lib.orc_debug_pixel.argtypes = [C.c_void_p, C.c_int64]
        o2.lib.orc_debug_rays.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float))]
        o2.lib.orc_debug_rays.restype = C.c_uint32
        o2.lib.orc_debug_pixel(o2.h, pix)
        o2.render(it + 1, threads=1)
        ptr = C.POINTER(C.c_float)()
        n = o2.lib.orc_debug_rays(o2.h, C.byref(ptr))
        rays = np.ctypeslib.as_array(ptr, shape=(n * 12,)).reshape(n, 12).copy()
        rays = rays[rays[:, 1] == it]
        for r in rays:
            if r[0] == 0:
                e, p, u, v, t = g.traceRays(r[2:5][None], r[5:8][None], r[8], r[9])
                oe, op, ou, ov, ot = o2.trace_closest(r[2:5][None], r[5:8][None], r[8], r[9])
                be, bp, bu, bv, bt = o2.trace_closest(r[2:5][None], r[5:8][None], r[8], r[9], brute=True)
                print("closest o=%s d=%s tmin=%g tmax=%g | gpu (%d,%d,t=%.9g) oracle (%d,%d,t=%.9g) brute (%d,%d,t=%.9g)" % (r[2:5], r[5:8], r[8], r[9], e[0].astype(np.int32), p[0].astype(np.int32), t[0], oe[0].astype(np.int32), op[0].astype(np.int32), ot[0], be[0].astype(np.int32), bp[0].astype(np.int32), bt[0]))
            else:
                occ = g.traceShadowRays(r[2:5][None], r[5:8][None], r[8], r[9])
                oo = o2.trace_any(r[2:5][None], r[5:8][None], r[8], r[9])
                bo = o2.trace_any(r[2:5][None], r[5:8][None], r[8], r[9], brute=True)
                print("shadow  o=%s d=%s dist=%.9g | gpu %d oracle %d brute %d" % (r[2:5], r[5:8], r[9], occ[0], oo[0], bo[0]))
        break
