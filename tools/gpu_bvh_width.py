"""Inner BVH records of four against six children (PRGPU_BVH_WIDTH) over a set of scenes: the builder's estimate of either tree, the width
`auto` takes, inner / leaf records per ray and ms per iteration under each forced width -- and whether `auto` took the faster one.
usage: python tools/gpu_bvh_width.py [scene ...]     (GPU box; scenes: c4 c5 soup100k soup4m box glassy rough sheets clusters spheres terrain; profiles/r05_bvh_width.log)"""
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


def cornell_with(pos, faces):
    """The Cornell box (camera, light, walls) with a mesh of its own inside instead of the uniform soup."""
    b = scene.SceneBuilder(W, H)
    b.settings.aa_sampler, b.settings.aa_samples = 0, 1024
    b.settings.aa_sampler = scene.abi.SAMPLER_SOBOL
    mats = scene._cornell_into(b)
    b.add_mesh(pos.astype(np.float32), faces.astype(np.uint32), mats["backWall"])
    return b.build()


def clusters(n=1_000_000, k=96, seed=3):
    """Triangles of uneven density: k Gaussian clusters whose radii span 1.5 decades, triangle size following the cluster's radius."""
    rng = np.random.default_rng(seed)
    centre = rng.uniform([-0.8, -0.8, 0.2], [0.8, 0.8, 1.7], (k, 3))
    radius = 10.0 ** rng.uniform(-2.3, -0.8, k)
    which = rng.integers(0, k, n)
    c = centre[which] + rng.normal(size=(n, 3)) * radius[which, None]
    c = np.clip(c, [-0.95, -0.95, 0.05], [0.95, 0.95, 1.9])
    size = 0.15 * radius[which, None]
    e1, e2 = rng.uniform(-1, 1, (n, 3)) * size, rng.uniform(-1, 1, (n, 3)) * size
    pos = np.stack([c, c + e1, c + e2], 1).reshape(-1, 3)
    return cornell_with(pos, np.arange(3 * n).reshape(n, 3))


def spheres(count=60, nlat=48, nlon=96, seed=5):
    """Tessellated spheres of different radii (closed regular meshes: what a modelled object looks like to the builder)."""
    rng = np.random.default_rng(seed)
    th, ph = np.linspace(0, np.pi, nlat + 1), np.linspace(0, 2 * np.pi, nlon, endpoint=False)
    unit = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)), np.outer(np.cos(th), np.ones_like(ph))], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nlat), np.arange(nlon), indexing="ij")
    a, b_, c, d = i * nlon + j, i * nlon + (j + 1) % nlon, (i + 1) * nlon + j, (i + 1) * nlon + (j + 1) % nlon
    quad = np.concatenate([np.stack([a, c, b_], -1).reshape(-1, 3), np.stack([b_, c, d], -1).reshape(-1, 3)])
    pos, faces = [], []
    for s_ in range(count):
        r = 10.0 ** rng.uniform(-1.7, -0.7)
        centre = rng.uniform([-0.8, -0.8, 0.2], [0.8, 0.8, 1.7])
        faces.append(quad + len(pos) * len(unit)); pos.append(unit * r + centre)
    return cornell_with(np.concatenate(pos), np.concatenate(faces))


def terrain(n=1000):
    """A height field of n x n cells (2 n^2 triangles) across the floor of the box."""
    x = np.linspace(-0.95, 0.95, n + 1)
    X, Y = np.meshgrid(x, x, indexing="ij")
    Z = 0.25 + 0.12 * np.sin(7 * X) * np.cos(5 * Y) + 0.04 * np.sin(31 * X + 17 * Y) + 0.01 * np.sin(113 * X) * np.sin(97 * Y)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a, b_, c, d = i * (n + 1) + j, i * (n + 1) + j + 1, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1
    return cornell_with(pos, np.concatenate([np.stack([a, c, b_], -1).reshape(-1, 3), np.stack([b_, c, d], -1).reshape(-1, 3)]))


SCENES = {"c4": lambda: scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000), "c5": c5,
          "soup100k": lambda: scene.cornell_soup(W, H, spp=1024, n_triangles=100_000), "soup4m": lambda: scene.cornell_soup(W, H, spp=1024, n_triangles=4_000_000),
          "box": lambda: scene.cornell_box(W, H, spp=1024), "glassy": lambda: scene.cornell_glassy(W, H, spp=1024), "rough": lambda: scene.cornell_rough(W, H, spp=1024),
          "sheets": sheets, "clusters": clusters, "spheres": spheres, "terrain": terrain}


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
    ra = run(sc, "auto")
    auto = ra[3]
    r4, r6 = run(sc, "4"), run(sc, "6")
    e4, e6 = auto["bvh_cost_4_wide"], auto["bvh_cost_6_wide"]
    best = min(r4[0], r6[0])
    right = ra[0] <= 1.01 * best
    print("%-9s   auto: %.3f ms, %.2f inner + %.2f leaf records per ray, %d-wide, top %d" % (name, ra[0], ra[1], ra[2], auto["bvh_width"], auto["bvh_top"]))
    print("%-9s | %7.2f / %7.2f (%.3f) | %5d | %8.3f %8.2f %8.2f   | %8.3f %8.2f %8.2f   | %.3f %.3f %s"
          % (name, e4, e6, e6 / max(e4, 1e-30), auto["bvh_width"], r4[0], r4[1], r4[2], r6[0], r6[1], r6[2], r6[1] / max(r4[1], 1e-30), r6[0] / r4[0],
             ("[top %d, stack bound %d]" % (auto["bvh_top"], auto["bvh_stack_bound"])) if right else ("<- auto is %.1f %% behind the better forced width" % (100.0 * (ra[0] / best - 1.0)))), flush=True)
