"""Strong-scaling probe on ONE GPU: the full C4 frame and every rank's tile share for world sizes 2, 4, 8 (the deal bench.py uses:
64x64 tiles up to two ranks, 16x16 beyond), ms per iteration, max over the ranks = what an N-GPU job would take before its reduce.
usage: python tools/gpu_shares.py [iterations] [worlds, e.g. 1,2,4,8] [--workload c4|c5]      (run on the GPU box under `timeout`; env knobs apply)
--workload c5: examples/complex.prc (BASELINE names 8 GPUs for it) instead of the 1M-triangle Cornell box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
workload = "c5" if "--workload" in sys.argv and sys.argv[sys.argv.index("--workload") + 1] == "c5" else "c4"
argv = [a for a in argv if a not in ("c4", "c5")]
iters = int(argv[0]) if len(argv) > 0 else 48
worlds = [int(a) for a in (argv[1] if len(argv) > 1 else "1,2,4,8").split(",")]
if workload == "c5":
    sc = scene.ArrayScene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = W, H
else:
    sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("PRGPU_"))
base = None
for world in worlds:
    times = []
    for rank in range(world):
        ctx = backend.RenderContext(sc)
        if world > 1:
            ctx.setTiles(tiling.tiles_for_rank(W, H, rank, world, tile=int(os.environ.get("SHARE_TILE", 64 if world <= 2 else 16))))   # SHARE_TILE: try another deal
        ctx.render(8); ctx.waitForFinish()          # warm-up (and the depth statistics tune_pixel_order wants)
        t = time.time(); ctx.render(iters); ctx.waitForFinish(); times.append((time.time() - t) / iters * 1e3)
        ctx.close()
    base = base or max(times)
    print("[%s %s] iters %3d  1/%d share: max %.3f ms/iteration (%.2fx) | %s" % (workload, tag, iters, world, max(times), base / max(times), " ".join("%.3f" % t for t in times)), flush=True)
