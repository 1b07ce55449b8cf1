"""The reference's own integrator-level test, `src/tests/python/whitefurnance.py`: a unit sphere of a white Lambert material inside
unit radiance, seen by an orthographic camera that also sees the background next to it -- every probed pixel must show the same
value (the reference probes nine points: centre, four inner, four corners that look past the sphere).  Parameters as there
(`:spectral_domain 520`, `direct`, `max_ray_depth 4`, hammersley, block filter radius 0, camera 2 x 2 at z = -1.0005; the full-spectrum
variants with D65 and `:spectral_hero true|false`).  The reference asserts 3 decimal places at 8 spp, which no Monte-Carlo estimator
with two MIS strategies can promise; here the probes are block means with more samples and the tolerance is stated.

The checker runs on the CPU; the HIP path renders the same scenes on the GPU (and is bit-identical to the checker)."""
import numpy as np
import pytest

from pearray_amd import scene
import oracle_binding as ob

POINTS = [(0.50, 0.50), (0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75), (0.05, 0.05), (0.95, 0.05), (0.05, 0.95), (0.95, 0.95)]   # whitefurnance.py:138-139


def furnace_scene(size, spp, mono=True, hero=True, nee=True):
    head = ":spectral_domain 520" if mono else ":spectral_hero %s" % ("true" if hero else "false")
    radiance = "1" if mono else "(illuminant 'D65')"
    albedo = "1" if mono else "'white'"
    spectral = "" if mono else "(sampler :slot 'spectral' :type 'random' :sample_count 1)"
    return scene.PrcScene(source="""(scene :name 'furnace' :render_width %d :render_height %d :camera 'Camera' %s
      (integrator :type 'DIRECT' :max_ray_depth 4 :nee %s)
      (sampler :slot 'aa' :type 'hammersley' :sample_count %d) %s
      (filter :slot 'pixel' :type 'BLOCK' :radius 0)
      (camera :name 'Camera' :type 'orthographic' :width 2 :height 2 :local_direction [0,0,1] :local_up [0,1,0] :local_right [1,0,0] :position [0,0,-1.0005])
      (light :name 'background' :type 'env' :radiance %s)
      (material :name 'Diffuse' :type 'diffuse' :albedo %s)
      (entity :type "sphere" :name "Unit Sphere" :radius 1 :material "Diffuse"))""" % (size, size, head, "true" if nee else "false", spp, spectral, radiance, albedo))


def predicted(fx, fy, nee_lost):
    """Radiance leaving the unit sphere towards the orthographic camera at film position (fx, fy), unit environment, white Lambert.
    The reference's constant environment samples the +z hemisphere only (cos_hemi, environment.cpp:91-93) but reports
    cos_hemi_pdf(|z|) for every direction (:75), so the balance heuristic gives the light strategy a share w_l = p_l / (p_l + p_b) of
    directions below the horizon that no light sample ever takes: L = 1 - int_{z<0} cos/pi w_l.  With nee_lost (one wavelength per
    sample: the NEE fragments are NaN and dropped, and the hero lane keeps 1/4 of the MIS denominator's four-lane sum,
    direct.cpp:440-441) what remains is 1/4 int cos/pi w_b."""
    x, y = 2 * fx - 1, 2 * fy - 1
    n = np.array([x, y, -np.sqrt(max(0.0, 1 - x * x - y * y))])
    th = (np.arange(240) + 0.5) / 240 * np.pi
    ph = (np.arange(480) + 0.5) / 480 * 2 * np.pi
    T, P = np.meshgrid(th, ph, indexing="ij")
    w = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], axis=-1)
    cos_n = np.maximum(0.0, w @ n)
    p_b, p_l = cos_n / np.pi, np.abs(w[..., 2]) / np.pi
    w_b = np.where(p_b + p_l > 0, p_b / np.maximum(p_b + p_l, 1e-30), 0.0)
    d_omega = np.sin(T) * (np.pi / 240) * (2 * np.pi / 480)
    if nee_lost:
        return 0.25 * float((p_b * w_b * d_omega).sum())
    return float((p_b * (w_b + (1 - w_b) * (w[..., 2] > 0)) * d_omega).sum())


def probes(xyz, size, half):
    out = []
    for fx, fy in POINTS:
        x, y = int(size * fx), int(size * fy)
        out.append(xyz[max(0, y - half):y + half + 1, max(0, x - half):x + half + 1, :].reshape(-1, 3).mean(axis=0))
    return np.array(out)


def check(xyz, size, feedback, nan_expected, noisy, nee):
    p = probes(xyz, size, half=size // 25)
    background = p[5:].mean(axis=0)                         # the corners look past the sphere
    assert np.allclose(p[5:], background, rtol=0.2 if noisy else 1e-5)   # constant; noise-free unless wavelengths are sampled
    yy, xx = np.mgrid[0:size, 0:size]
    on_sphere = ((xx + 0.5) / size * 2 - 1) ** 2 + ((yy + 0.5) / size * 2 - 1) ** 2 < 0.85   # whole pixels on the sphere
    mean = xyz[on_sphere].mean(axis=0)
    r2 = ((xx + 0.5) / size * 2 - 1) ** 2 + ((yy + 0.5) / size * 2 - 1) ** 2
    white = np.ones(3) if not noisy else np.array([0.9505, 1.0, 1.0891])      # white under D65 is the D65 white point
    if nan_expected:
        # One wavelength per sample + next event estimation of a non-delta light: the MIS weight divides by the hero mask (1,0,0,0)
        # (direct.cpp:321), the camera importance was multiplied by the same mask (RenderTile.cpp:127-128): inf * 0 = NaN in
        # OutputSpectralEntry::contribution, and the output device drops the fragment with the NaN feedback bit
        # (LocalFrameOutputDevice.cpp:127-141).  The reference's test_spec / test_non_hero expectation (sphere == 1) is NOT met by the
        # reference code as it stands; this restatement keeps the code's behaviour and predicts its value.
        assert (feedback[(r2 > 0.3) & (r2 < 0.9)] & 1).all()      # near the pole every light sample lies below the surface: no fragment at all
        assert not feedback[r2 > 1.15].any()
    elif feedback is not None:
        assert not feedback.any()
    if not nee:                                                    # the plain furnace: BSDF sampling alone sees unit radiance everywhere
        for k in range(5):
            assert np.allclose(p[k], white, rtol=0.1 if noisy else 1e-5), (k, p[k])
        assert np.allclose(mean, white, rtol=0.03 if noisy else 1e-5), mean   # 390-830 nm of D65, upsampled white: within 3 % of the tabulated white point
        if not noisy:
            assert np.allclose(background, 1.0, rtol=1e-6)         # mono: sphere == background == 1, the reference's expectation
        return
    for k in range(5):
        want = predicted(*POINTS[k], nee_lost=nan_expected) * white
        assert np.allclose(p[k], want, rtol=0.12 if noisy else 0.04), (k, p[k], want)
    grid = [((i + 0.5) / 24, (j + 0.5) / 24) for i in range(24) for j in range(12)]      # half the disc: the value is symmetric in y
    inside = [predicted(fx, fy, nee_lost=nan_expected) for fx, fy in grid if (2 * fx - 1) ** 2 + (2 * fy - 1) ** 2 < 0.85]
    assert np.allclose(mean, np.mean(inside) * white, rtol=0.04 if noisy else 0.01), (mean, np.mean(inside))


CASES = [("spec", True, True, True), ("spec_no_nee", True, True, False), ("non_hero", False, False, True), ("non_hero_no_nee", False, False, False),
         ("full", False, True, True), ("full_no_nee", False, True, False)]   # name, mono, hero, nee


def expectations(mono, hero, nee):
    # One wavelength per sample (`:spectral_domain`, or `:spectral_hero false`: both force IsMonochrome, RenderTile.cpp:124-128)
    single = mono or not hero
    return dict(nan_expected=single and nee, noisy=not mono, nee=nee)


@pytest.mark.parametrize("name,mono,hero,nee", CASES)
def test_checker_white_furnace(name, mono, hero, nee):
    size, spp = 50, 128
    sc = furnace_scene(size, spp, mono=mono, hero=hero, nee=nee)
    o = ob.OracleScene(sc)
    o.render(spp, threads=8)
    xyz, _, fb = o.output()
    if mono:
        assert np.array_equal(xyz[..., 0], xyz[..., 1]) and np.array_equal(xyz[..., 0], xyz[..., 2])   # the raw spectral value in every channel
    check(xyz, size, feedback=fb, **expectations(mono, hero, nee))


@pytest.mark.gpu
@pytest.mark.parametrize("name,mono,hero,nee", CASES)
def test_gpu_white_furnace_at_the_reference_size(name, mono, hero, nee):
    from pearray_amd import backend
    size, spp = 200, 128                                    # IMGSIZE = 200 (whitefurnance.py:9)
    sc = furnace_scene(size, spp, mono=mono, hero=hero, nee=nee)
    g = backend.RenderContext(sc)
    g.start(); g.waitForFinish()
    xyz, _, fb = g.output()
    check(xyz, size, feedback=fb, **expectations(mono, hero, nee))
    small = furnace_scene(40, 8, mono=mono, hero=hero, nee=nee)                      # the reference's 8 spp, against the checker bit for bit
    g2 = backend.RenderContext(small); g2.start(); g2.waitForFinish()
    o = ob.OracleScene(small); o.render(8, threads=8)
    assert np.array_equal(g2.output()[0], o.output()[0]) and np.array_equal(g2.output()[2], o.output()[2])
