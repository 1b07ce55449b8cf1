"""The sorted-ray-queue experiment (north_star: "sorted ray queues for coalesced HBM reads"): the lockstep pipeline of the C4 frame with and
without PRGPU_SORT_RAYS=1 (each path depth's ray list radix-sorted by Morton(origin) | direction octant before it is traced), against
the persistent pipeline.  Prints ms per iteration, the traversal / sort kernel times and checks that the frames are identical.
usage: python tools/gpu_sorted_rays.py [iterations] [render-only]       (GPU box; `render-only`: the program rocprofv3 wraps)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2 and sys.argv[2] == "render-only":
    from pearray_amd import backend, scene
    ctx = backend.RenderContext(scene.cornell_soup(1920, 1080, spp=1024, n_triangles=1_000_000))
    ctx.render(iters); ctx.waitForFinish()
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np
    from pearray_amd import backend, scene
    ctx = backend.RenderContext(scene.cornell_soup(1920, 1080, spp=1024, n_triangles=1_000_000))
    ctx.render(2); ctx.waitForFinish()
    ctx.setTiming(True)
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = (time.time() - t) / iters * 1e3
    fam = {k: ctx.kernelTime(k) for k in ("trace_closest", "trace_any", "shade", "sort", "raygen", "resolve", "path")}
    xyz, smp, fb = ctx.output()
    import hashlib
    print(json.dumps({"ms_per_iteration": dt, "families_ms_per_iteration": {k: v[0] / iters for k, v in fam.items() if v[1]},
                      "launches_per_iteration": {k: v[1] / iters for k, v in fam.items() if v[1]}, "frame_sha": hashlib.sha256(xyz.tobytes()).hexdigest()[:16]}))
    sys.exit(0)
out = {}
for name, env in (("lockstep", {"PRGPU_MODE": "lockstep"}), ("lockstep_sorted", {"PRGPU_MODE": "lockstep", "PRGPU_SORT_RAYS": "1"}), ("persistent", {})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(iters), "child"], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out[name] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
    print(name, out[name], flush=True)
same = len({v.get("frame_sha") for v in out.values()}) == 1
print(json.dumps({"iterations": iters, "frames_identical": same, **out}))
