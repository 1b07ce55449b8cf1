"""Probe: lockstep vs persistent render mode -- identical output?  how fast?  (run on the GPU box under `timeout`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pearray_amd import backend, scene, tiling

def run(mode, sc, iters, tiles=None, warm=0):
    os.environ["PRGPU_MODE"] = mode
    ctx = backend.RenderContext(sc)
    if tiles:
        ctx.setTiles(tiles)
    if warm:
        ctx.render(warm); ctx.waitForFinish()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
    out = ctx.output(); st = ctx.statistics(); ph = ctx.primaryHits()
    ctx.close()
    return out, st, ph, dt

def compare(name, sc, iters, tiles=None, warm=0):
    a = run("lockstep", sc, iters, tiles, warm)
    b = run("persistent", sc, iters, tiles, warm)
    same = all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    nbad = int((a[0][0] != b[0][0]).any(axis=-1).sum())
    print("%-28s identical=%s (pixels differing %d) stats_equal=%s  lockstep %.2f ms/iter  persistent %.2f ms/iter" % (
        name, same, nbad, a[1] == b[1], a[3] / iters * 1e3, b[3] / iters * 1e3), flush=True)
    if a[1] != b[1]:
        print("   ", a[1], "\n   ", b[1])
    return same

which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    compare("cornell 64x64 x4", scene.cornell_box(64, 64, spp=4), 4)
    compare("cornell 256x256 x16", scene.cornell_box(256, 256, spp=16), 16)
    compare("soup20k 320x200 x8", scene.cornell_soup(320, 200, spp=8, n_triangles=20_000), 8)
    compare("sphere 128x128 x8", scene.sphere_light(128, 128, spp=8), 8)
elif which == "filters":
    from pearray_amd import _cabi as abi
    W, H = 1920, 1080
    for flt, r, name in ((abi.FILTER_GAUSSIAN, 2, "gaussian r=2"), (abi.FILTER_MITCHELL, 3, "mitchell r=3"), (abi.FILTER_TRIANGLE, 1, "triangle r=1")):
        compare("C4 full frame, %s x16" % name, scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000, filter=flt, filter_radius=r), 16, warm=2)
else:
    W, H = 1920, 1080
    sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
    compare("C4 full frame x16", sc, 16, warm=2)
    for world in (2, 4, 8):
        compare("C4 rank0 of %d x16" % world, sc, 16, tiles=tiling.tiles_for_rank(W, H, 0, world), warm=2)
