"""Where does a C5 shading pass's time go?  The same frame with next event estimation switched off (another image: diagnosis only), per-queue
pass times of the instrumented kernel variant.  usage: PRGPU_DEBUG_COUNTERS=1 python tools/probe_c5_nee.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import backend, scene
for nee in (1, 0):
    sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = 1920, 1080
    sc.desc.settings.nee = nee
    ctx = backend.RenderContext(sc)
    ctx.render(8); ctx.waitForFinish()
    t = time.time(); ctx.render(16); ctx.waitForFinish(); dt = (time.time() - t) / 16 * 1e3
    sys.stderr.write("---- nee = %d: %.2f ms per iteration (plain kernel); instrumented time split:\n" % (nee, dt)); sys.stderr.flush()
    ctx.setInstrumentation(True); ctx.render(8); ctx.waitForFinish(); ctx.setInstrumentation(False)
    ctx.traceCounters()
    ctx.close()
