"""What bounds a small tile share: rank 0's 1/8 share of the C4 frame at several iterations per render call, with the trace
counters of an instrumented pass next to the timing.  usage: [TILE=16] python tools/gpu_probe_share8.py [world] [iterations ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
for iters in [int(a) for a in sys.argv[2:]] or [8, 32, 96, 256]:
    ctx = backend.RenderContext(sc)
    if world > 1:
        ctx.setTiles(tiling.tiles_for_rank(W, H, int(os.environ.get("RANK_PROBE", "0")), world, tile=int(os.environ.get("TILE", "64"))))
    ctx.render(8); ctx.waitForFinish()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = (time.time() - t) / iters * 1e3
    tc0 = ctx.traceCounters(); ctx.setInstrumentation(True); ctx.render(iters); ctx.waitForFinish(); ctx.setInstrumentation(False); tc1 = ctx.traceCounters()
    d = {k: tc1[k] - tc0[k] for k in tc1}
    steps = d["wave_steps_closest"]
    leaf_batches = d["wave_steps_any"]   # split traversal: leaf batches, counted in the steps too
    recs = d["nodes_closest"] + d["leaves_closest"] + d["nodes_any"] + d["leaves_any"]
    print("share 1/%d  iters %3d  %.3f ms/iteration | lane utilisation %.3f, wave steps per iteration %.0f, shading %.1f %% / idle %.1f %% of wave time, pass fill %.2f%s, mean wave lifetime %.3f ms/iteration (100 MHz ticks, 3072 waves)" % (
        world, iters, dt, recs / max(64 * steps, 1), steps / iters, 100.0 * d["shade_ticks"] / max(d["total_ticks"], 1), 100.0 * d["idle_ticks"] / max(d["total_ticks"], 1), d["shade_lanes"] / max(64 * d["shade_batches"], 1),
        (", leaf batches %.0f per iteration of %.1f tasks" % (leaf_batches / iters, (d["leaves_closest"] + d["leaves_any"]) / leaf_batches)) if leaf_batches else "", d["total_ticks"] / 3072 / 1e5 / iters), flush=True)
    ctx.close()
