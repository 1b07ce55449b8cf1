"""Sweep the persistent kernel's knobs on the C4 workload (full frame and rank 0 of 8)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling
W, H = 1920, 1080
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
iters = int(os.environ.get("SWEEP_ITERS", "48"))

def run(env, world, instrument=False):
    for k, v in env.items():
        os.environ[k] = str(v)
    ctx = backend.RenderContext(sc)
    if world > 1:
        ctx.setTiles(tiling.tiles_for_rank(W, H, 0, world))
    ctx.render(2); ctx.waitForFinish()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
    extra = ""
    if instrument:
        tc0 = ctx.traceCounters()
        ctx.setInstrumentation(True); ctx.render(2); ctx.waitForFinish(); ctx.setInstrumentation(False)
        tc = ctx.traceCounters()
        recs = tc["nodes_closest"] + tc["leaves_closest"] + tc["nodes_any"] + tc["leaves_any"]
        steps = tc["wave_steps_closest"] + tc["wave_steps_any"]
        extra = " lane-util %.3f shade-util %.3f (batches %d, %.1f us/pass) time: shade %.2f idle %.2f trace %.2f; %.2f us/step" % (
            recs / max(64 * steps, 1), tc["shade_lanes"] / max(64 * tc["shade_batches"], 1), tc["shade_batches"],
            tc["shade_ticks"] / max(tc["shade_batches"], 1) / 100.0, tc["shade_ticks"] / max(tc["total_ticks"], 1), tc["idle_ticks"] / max(tc["total_ticks"], 1),
            1 - (tc["shade_ticks"] + tc["idle_ticks"]) / max(tc["total_ticks"], 1),
            (tc["total_ticks"] - tc["shade_ticks"] - tc["idle_ticks"]) / max(steps, 1) / 100.0)
    ctx.close()
    return dt / iters * 1e3, extra

base = {"PRGPU_MODE": "persistent", "PRGPU_PP_BLOCKS_PER_CU": 3, "PRGPU_PP_REFILL": 48, "PRGPU_PP_SLOTS": 512, "PRGPU_PP_SHADE_MIN": 64, "PRGPU_PP_SHADE_PARTIAL": 16, "PRGPU_PP_PARTIAL_ACT": 64, "PRGPU_PP_REFILL_MIN": 1, "PRGPU_PP_BOTH": 0, "PRGPU_PP_OCCUPANCY": 3}
configs = [dict(base)]
for name in sys.argv[1:]:
    c = dict(base)
    for kv in name.split("+"):   # KEY=v[+KEY2=v2...] is one configuration
        k, v = kv.split("=")
        c[k] = v
    configs.append(c)
for c in configs:
    r1, e1 = (run(c, 1, True) if not os.environ.get("SWEEP_SKIP_FULL") else (0.0, ""))
    world = int(os.environ.get("SWEEP_WORLD", "8"))
    r8, e8 = run(c, world, True)
    print({k[6:]: v for k, v in c.items() if k != "PRGPU_MODE"}, "full %.2f ms/iter%s | 1/%d share %.2f ms/iter%s -> %.2fx" % (r1, e1, world, r8, e8, r1 / r8), flush=True)
