"""Inner BVH records of four against six children (PRGPU_BVH_WIDTH) over a set of scenes: the builder's estimate of either tree, the width
`auto` takes, inner / leaf records per ray and ms per iteration under each forced width -- and whether `auto` took the faster one.
usage: python tools/gpu_bvh_width.py [scene ...]     (GPU box; scenes: c4 c5 soup100k soup4m box glassy rough sheets; profiles/r05_bvh_width.log)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pearray_amd import backend, scene

W, H = 1920, 1080


def sheets():
    n = 16384
    b = scene.SceneBuilder(W // 4, H // 4)
    b.settings.aa_samples = 16
    z = np.arange(n, dtype=np.float32) * 0.001
    tri = np.array([[0, 0], [2, 0], [0, 2]], dtype=np.float32)
    pos = np.concatenate([np.repeat(tri[None], n, 0), np.repeat(z[:, None, None], 3, 1)], 2).reshape(-1, 3).astype(np.float32)
    b.add_mesh(pos, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3), b.lambert(b.spectrum_const(0.7)))
    light = np.array([[0.5, 0.5, -2.0], [1.5, 0.5, -2.0], [0.5, 1.5, -2.0]], dtype=np.float32)
    b.add_mesh(light, np.array([[0, 1, 2]], dtype=np.uint32), b.lambert(b.spectrum_const(0.0)), emission=b.diffuse_emission(b.illuminant_d65()))
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.6, 0.6, -3.0)
    b.set_camera(T, width=1.6, height=0.9, ortho=True)
    return b.build()


def c5():
    sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = W, H
    return sc


SCENES = {"c4": lambda: scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000), "c5": c5,
          "soup100k": lambda: scene.cornell_soup(W, H, spp=1024, n_triangles=100_000), "soup4m": lambda: scene.cornell_soup(W, H, spp=1024, n_triangles=4_000_000),
          "box": lambda: scene.cornell_box(W, H, spp=1024), "glassy": lambda: scene.cornell_glassy(W, H, spp=1024), "rough": lambda: scene.cornell_rough(W, H, spp=1024),
          "sheets": sheets}


def run(sc, width, iters=16):
    os.environ["PRGPU_BVH_WIDTH"] = width
    ctx = backend.RenderContext(sc)
    ctx.render(8); ctx.waitForFinish()
    best = 1e30
    for _ in range(2):
        t = time.time(); ctx.render(iters); ctx.waitForFinish()
        best = min(best, (time.time() - t) / iters * 1e3)
    a = ctx.traceCounters()
    ctx.setInstrumentation(True); ctx.render(4); ctx.waitForFinish(); ctx.setInstrumentation(False)
    b = ctx.traceCounters()
    rays = max(b["rays_closest"] + b["rays_any"] - a["rays_closest"] - a["rays_any"], 1)
    inner = (b["nodes_closest"] + b["nodes_any"] - a["nodes_closest"] - a["nodes_any"]) / rays
    leaf = (b["leaves_closest"] + b["leaves_any"] - a["leaves_closest"] - a["leaves_any"]) / rays
    info = ctx.pipelineInfo()
    ctx.close()
    return best, inner, leaf, info


print("%-9s | %22s | %5s | %28s | %28s | %s" % ("scene", "estimate 4 / 6 (ratio)", "auto", "4-wide: ms, inner, leaf / ray", "6-wide: ms, inner, leaf / ray", "6 against 4: records, time"))
for name in (sys.argv[1:] or list(SCENES)):
    sc = SCENES[name]()
    _, _, _, auto = run(sc, "auto", iters=2)
    r4, r6 = run(sc, "4"), run(sc, "6")
    e4, e6 = auto["bvh_cost_4_wide"], auto["bvh_cost_6_wide"]
    right = (auto["bvh_width"] == 6) == (r6[0] < r4[0])
    print("%-9s | %7.2f / %7.2f (%.3f) | %5d | %8.3f %8.2f %8.2f   | %8.3f %8.2f %8.2f   | %.3f %.3f %s"
          % (name, e4, e6, e6 / max(e4, 1e-30), auto["bvh_width"], r4[0], r4[1], r4[2], r6[0], r6[1], r6[2], r6[1] / max(r4[1], 1e-30), r6[0] / r4[0],
             ("[stack bound %d]" % auto["bvh_stack_bound"]) if right else ("<- auto took the slower tree (%.1f %%)" % (100.0 * abs(r6[0] / r4[0] - 1.0)))), flush=True)
