"""Small films on one GPU with either organisation of the persistent kernel: ms per iteration of the Cornell box (variant 1 of the kernel) and
of the 1M-triangle scene at film sizes from 128^2 to 1024^2 -- where does the latency organisation (PRGPU_PP_KERNEL=latency) pay?
usage: python tools/gpu_small_films.py   (GPU box; run once per PRGPU_PP_KERNEL value)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearray_amd import backend, scene

tag = os.environ.get("PRGPU_PP_KERNEL", "auto")
for name, make in (("cornell", lambda w: scene.cornell_box(w, w, spp=256)), ("soup1M", lambda w: scene.cornell_soup(w, w, spp=256, n_triangles=1_000_000))):
    for w in (128, 256, 384, 512, 768, 1024):
        ctx = backend.RenderContext(make(w))
        ctx.render(8); ctx.waitForFinish()
        t = time.time(); ctx.render(64); ctx.waitForFinish(); dt = (time.time() - t) / 64 * 1e3
        info = ctx.pipelineInfo()
        print("[%s] %-8s %4dx%-4d %7.0f k pixels: %.3f ms/iteration  (%s kernel, %d blocks x %d slots)" % (tag, name, w, w, w * w / 1e3, dt, info["kernel"], info["blocks"], info["slots_per_block"]), flush=True)
        ctx.close()
