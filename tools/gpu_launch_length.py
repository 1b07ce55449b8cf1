"""Cost of a bounded launch: ms per iteration of render calls of n iterations (one persistent launch each), full C4 (or C5) frame.
T(n) = fixed + n * per_iteration: the fixed part is the fill and the drain of the launch.
usage: python tools/gpu_launch_length.py [scene: c4|c5] [lengths, e.g. 4,8,20,48,96]   (GPU box; env knobs apply)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene

W, H = 1920, 1080
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
lengths = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "4,8,20,48,96").split(",")]
if which == "c5":
    sc = scene.ArrayScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = W, H
else:
    sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
ctx.render(8); ctx.waitForFinish()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("PRGPU_"))
xs, ys = [], []
for n in lengths:
    best = 1e9
    for rep in range(2):
        t = time.perf_counter(); ctx.render(n); ctx.waitForFinish(); best = min(best, (time.perf_counter() - t) * 1e3)
    xs.append(n); ys.append(best)
    print("[%s] %s render(%3d): %8.2f ms = %.3f ms/iteration" % (tag, which, n, best, best / n), flush=True)
b, a = np.polyfit(xs, ys, 1)
print("[%s] fit: %.2f ms fixed per launch + %.3f ms per iteration" % (tag, a, b))
