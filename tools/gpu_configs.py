"""BASELINE.json configs C1-C3 (and C4 at reduced iteration count) on one GPU, each checked against the CPU oracle at FULL
resolution for a few iterations and then timed over more iterations.  Prints one line per config.
usage: python tools/gpu_configs.py   (on the GPU box, under `timeout`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_binding import OracleScene
from pearray_amd import _cabi as abi
from pearray_amd import backend, scene

def run(name, sc, check_iters, time_iters):
    ctx = backend.RenderContext(sc)
    ctx.render(check_iters); ctx.waitForFinish()
    xyz, smp, fb = ctx.output()
    ge, gp = ctx.primaryHits()
    ora = OracleScene(sc)
    t = time.time(); ora.render(check_iters, threads=os.cpu_count()); t_cpu = time.time() - t
    oxyz, osmp, ofb = ora.output()
    oe, op = ora.primary_hits()
    ids = bool(np.array_equal(ge, oe) and np.array_equal(gp, op))
    exact = bool(np.array_equal(xyz, oxyz) and np.array_equal(smp, osmp) and np.array_equal(fb, ofb))
    rel = float(np.sqrt(((xyz.astype(np.float64) - oxyz) ** 2).sum()) / max(np.sqrt((oxyz.astype(np.float64) ** 2).sum()), 1e-30))
    s0 = ctx.statistics()
    t = time.time(); ctx.render(time_iters); ctx.waitForFinish(); dt = time.time() - t
    s1 = ctx.statistics()
    n = s1["pixel_samples"] - s0["pixel_samples"]
    rays = sum(s1[k] - s0[k] for k in ("primary_rays", "bounce_rays", "shadow_rays"))
    px = sc.width * sc.height
    print("%-44s hit ids equal=%s frame bit-exact=%s rel_l2=%.2e | GPU %.1f Msamples/s %.0f Mrays/s (%d it, %.2f ms/it) | oracle %.3f Msamples/s on %d threads"
          % (name, ids, exact, rel, n / dt / 1e6, rays / dt / 1e6, time_iters, dt / time_iters * 1e3, px * check_iters / t_cpu / 1e6, os.cpu_count()), flush=True)
    ctx.close()

run("C1 cornellbox 256x256 mjitt 16 spp", scene.cornell_box(256, 256, spp=16), 4, 12)
run("C2 sphere + area light 512x512 mjitt 64 spp", scene.sphere_light(512, 512, spp=64), 4, 60)
run("C3 cornellbox 1024x1024 mjitt 256 spp", scene.cornell_box(1024, 1024, spp=256), 2, 128)
run("C3b evaluation scene 256x256 sobol 128 spp", scene.cbox_eval(256, 256, spp=128), 4, 124)
run("C4 1M-triangle cornell 1920x1080 sobol", scene.cornell_soup(1920, 1080, spp=1024, n_triangles=1_000_000), 1, 32)
