"""What the host pays around the timed region of bench.py (C4): scene upload + BVH build, the frame's download, and the rate of a whole
job of K iterations with both included.  usage: python tools/probe_host_inclusive.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = scene.cornell_soup(1920, 1080, spp=1024, n_triangles=1_000_000)
for attempt in range(2):                     # the second pass is the warm one (library, code objects and allocator already there)
    t0 = time.perf_counter(); g = backend.RenderContext(sc); t1 = time.perf_counter()
    g.render(K); g.waitForFinish(); t2 = time.perf_counter()
    xyz, smp, fb = g.output(); t3 = time.perf_counter()
    n = 1920 * 1080 * K
    print("pass %d: scene create (upload %d triangles + LBVH + tables) %.3f s, %d iterations %.3f s (first launch = calibration), download of %.1f MB %.4f s"
          " -> %.1f Msamples/s render only, %.1f with the download, %.1f with scene create and download"
          % (attempt, sc.desc.n_triangles, t1 - t0, K, t2 - t1, (xyz.nbytes + smp.nbytes + fb.nbytes) / 1e6, t3 - t2, n / (t2 - t1) / 1e6, n / (t3 - t1) / 1e6, n / (t3 - t0) / 1e6))
    g.close()
