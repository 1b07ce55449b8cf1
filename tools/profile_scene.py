"""Render N iterations of a named scene in one launch of the plain kernel, after a warm-up of 8 (no checks, no instrumentation): the program rocprofv3 wraps for the scenes bench.py
does not cover.  usage: python3 tools/profile_scene.py c5|rough|metal_all|share8 N   (share8 = rank 0's tiles of the C4 frame at 8 ranks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import backend, scene

name, iters = sys.argv[1], int(sys.argv[2])
if name == "c5":
    sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))
elif name == "rough":
    sc = scene.cornell_rough(1024, 1024, spp=256, roughness=0.2, vndf=True)
elif name == "metal_all":
    os.environ["PRGPU_FORCE_FEATURES"] = "255"
    sc = scene.cornell_metal(1024, 1024, spp=256)
elif name.startswith("share"):
    sc = scene.cornell_soup(1920, 1080, spp=1024, n_triangles=1_000_000)
else:
    raise SystemExit("unknown scene")
ctx = backend.RenderContext(sc)
if name.startswith("share"):
    from pearray_amd import tiling
    ctx.setTiles(tiling.tiles_for_rank(1920, 1080, 0, int(name[5:])))
ctx.render(8)          # the scene's first launch is its calibration launch (instrumented kernel variant, prgpu_api.hip render_persistent)
ctx.waitForFinish()
ctx.render(iters)      # ... the profiled one: the plain kernel, `iters` iterations in one launch
ctx.waitForFinish()
print("rendered %d iterations of %s: %s" % (iters, name, ctx.statistics()))
