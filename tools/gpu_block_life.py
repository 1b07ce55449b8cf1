"""Distribution of block lifetimes of one instrumented persistent launch (PRGPU_DUMP_BLOCK_LIFE diagnostics).
usage: python tools/gpu_block_life.py [world] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "block_life.txt")
os.environ["PRGPU_DUMP_BLOCK_LIFE"] = out
from pearray_amd import backend, scene, tiling

W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sc = scene.cornell_soup(W, H, spp=1024, n_triangles=1_000_000)
ctx = backend.RenderContext(sc)
if world > 1:
    ctx.setTiles(tiling.tiles_for_rank(W, H, 0, world, tile=64 if world <= 2 else 16))
ctx.render(8); ctx.waitForFinish()
ctx.setInstrumentation(True); ctx.render(iters); ctx.waitForFinish()
rows = np.loadtxt(out, dtype=np.int64)
life = rows[:, 1] / 1e5 / iters     # ms per iteration
work = rows[:, 2] / iters           # path vertices per iteration
q = np.percentile(life, [0, 5, 25, 50, 75, 95, 100])
print("share 1/%d, %d iterations: block lifetime ms/iteration min %.3f p5 %.3f p25 %.3f median %.3f p75 %.3f p95 %.3f max %.3f, mean %.3f" % ((world, iters) + tuple(q) + (life.mean(),)))
print("launch: last block ends after %.3f ms; blocks idle at the end of the launch for %.1f %% of the grid's time (1 - mean / max lifetime)" % (rows[:, 1].max() / 1e5, 100 * (1 - life.mean() / life.max())))
print("vertices per block and iteration: min %.0f median %.0f max %.0f; correlation(lifetime, vertices) = %.3f" % (work.min(), np.median(work), work.max(), np.corrcoef(life, work)[0, 1]))
for k in range(8):  # by XCD (block index mod 8)
    print("  blocks with index %% 8 == %d: mean lifetime %.3f, mean vertices %.0f" % (k, life[k::8].mean(), work[k::8].mean()))
order = np.argsort(life)
print("slowest blocks:", [(int(rows[i, 0]), round(float(life[i]), 3), int(work[i])) for i in order[-6:]])
print("fastest blocks:", [(int(rows[i, 0]), round(float(life[i]), 3), int(work[i])) for i in order[:6]])
cost = ctx.pathCost().astype(np.float64).reshape(-1)
own = cost > 0
if not own.any():   # the per-pixel path cost is only kept while every owned pixel is in flight at once (small tile shares)
    sys.exit(0)
c = cost[own] / (8 + iters)      # vertices per sample (the warm-up iterations count too)
inbox = c > 1.5
print("owned pixels %d, in-box %d; vertices per sample: mean %.2f, in-box mean %.2f p50 %.2f p90 %.2f p99 %.2f p99.9 %.2f max %.2f" % (
    own.sum(), inbox.sum(), c.mean(), c[inbox].mean(), *np.percentile(c[inbox], [50, 90, 99, 99.9]), c.max()))
