"""Throughput of the FULL kernel variant (delta / rough / principled materials, infinite lights, spheres) next to the lean one on
Cornell-box scenes at 1024x1024 -- the state of the C5 feature set on one GPU.  usage: python tools/gpu_full_variant.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import _cabi as abi
from pearray_amd import backend, scene

def run(name, sc, iters=64):
    ctx = backend.RenderContext(sc)
    ctx.render(4); ctx.waitForFinish()
    s0 = ctx.statistics()
    t = time.time(); ctx.render(iters); ctx.waitForFinish(); dt = time.time() - t
    s1 = ctx.statistics()
    n = s1["pixel_samples"] - s0["pixel_samples"]
    rays = sum(s1[k] - s0[k] for k in ("primary_rays", "bounce_rays", "shadow_rays"))
    depth = (s1["camera_depth"] - s0["camera_depth"]) / max(n, 1)
    tc0 = ctx.traceCounters(); ctx.setInstrumentation(True); ctx.render(4); ctx.waitForFinish(); ctx.setInstrumentation(False); tc1 = ctx.traceCounters()
    d = {k: tc1[k] - tc0[k] for k in tc1 if isinstance(tc1[k], int)}
    print("%-52s %7.1f Msamples/s %7.0f Mrays/s  mean depth %.2f  %.2f ms/iteration | shading %4.1f %% of wave time, pass fill %.2f" % (
        name, n / dt / 1e6, rays / dt / 1e6, depth, dt / iters * 1e3, 100.0 * d["shade_ticks"] / max(d["total_ticks"], 1), d["shade_lanes"] / max(64 * d["shade_batches"], 1)), flush=True)
    ctx.close()

W = H = 1024
run("lambert cornell (lean variant)", scene.cornell_box(W, H, spp=256))
run("glass boxes (bk7)", scene.cornell_glassy(W, H, spp=256))
run("metal boxes", scene.cornell_metal(W, H, spp=256))
def metal_env():
    b = scene.SceneBuilder(W, H)
    b.settings.aa_sampler, b.settings.aa_samples = abi.SAMPLER_MJITT, 256
    scene._cornell_into(b, material_override={"tallBox": lambda bb: bb.conductor(), "shortBox": lambda bb: bb.conductor()})
    b.environment_light(b.spectrum_const(0.2))
    return b.build()
run("metal boxes + environment light (no-rough variant)", metal_env())
run("rough conductor/dielectric boxes, vndf", scene.cornell_rough(W, H, spp=256, roughness=0.2, vndf=True))
os.environ["PRGPU_FORCE_FEATURES"] = "255"
run("metal boxes, ALL-FEATURES kernel forced", scene.cornell_metal(W, H, spp=256))
run("lambert cornell, ALL-FEATURES kernel forced", scene.cornell_box(W, H, spp=256))
del os.environ["PRGPU_FORCE_FEATURES"]
os.environ["PRGPU_PP_OCCUPANCY"] = "2"
run("rough boxes, 2 waves per SIMD (256 VGPRs)", scene.cornell_rough(W, H, spp=256, roughness=0.2, vndf=True))
del os.environ["PRGPU_PP_OCCUPANCY"]
