"""prcmp statistics (src/tools/imgcmp/main.cpp:151-330) through prgpu_image_compare / prgpu_image_stats_merge and tools/imgcmp.py:
closed forms in numpy, the crop clamping of the reference, Inf / NaN handling, and the tool end to end on EXR files written by the
library's own writer.  CPU only."""
import io
import os
from contextlib import redirect_stdout

import numpy as np

from pearray_amd import backend

import imgcmp


def planes(seed=3, shape=(37, 53)):
    rng = np.random.default_rng(seed)
    ref = rng.uniform(0.0, 2.0, shape).astype(np.float32)
    img = (ref + rng.normal(0.0, 0.05, shape)).astype(np.float32)
    return img, ref


def test_statistics_follow_the_closed_forms():
    img, ref = planes()
    st = backend.image_compare(img, ref)
    a, b = img.astype(np.float64), ref.astype(np.float64)
    d = np.abs(a - b)
    assert st.n == img.size and st.inf_count == 0 and st.nan_count == 0
    assert st.min == img.min() and st.max == img.max() and st.min_ref == ref.min() and st.max_ref == ref.max()
    assert st.max_diff == np.abs(img - ref).max() and st.min_diff == np.abs(img - ref).min()
    for got, want in ((st.mean, a.mean()), (st.mean_ref, b.mean()), (st.mean_sqr, (a * a).mean()), (st.mean_sqr_ref, (b * b).mean()),
                      (st.mean_diff, d.mean()), (st.mse, (d * d).mean()), (st.mape, (d / np.abs(b)).mean())):
        assert abs(got - want) <= 2e-5 * max(abs(want), 1e-3), (got, want)   # fp32 running sums like the reference
    rep = dict(backend.image_stats_report(st))
    assert abs(rep["RMSE"] - np.sqrt((d * d).mean())) < 1e-5 and rep["MAE"] == st.mean_diff
    assert rep["PSNR"].endswith("dB]") and abs(rep["Variance"] - a.var()) < 1e-4


def test_crop_is_clamped_like_the_reference_and_nonfinite_pixels_are_counted():
    img, ref = planes()
    st = backend.image_compare(img, ref, crop=(5, 7, 20, 30))
    sub_a, sub_b = img[7:30, 5:20], ref[7:30, 5:20]
    assert st.n == sub_a.size and st.max == sub_a.max() and abs(st.mean_ref - sub_b.astype(np.float64).mean()) < 1e-5
    st = backend.image_compare(img, ref, crop=(500, 500, 600, 600))   # outside: the last pixel remains (main.cpp:297-300)
    assert st.n == 1 and st.max == img[-1, -1]
    st = backend.image_compare(img, ref, crop=(10, 10, 5, 5))         # inverted: one pixel at the start corner
    assert st.n == 1 and st.max == img[10, 10]
    bad = img.copy()
    bad[0, 0], bad[1, 1], bad[2, 2] = np.inf, np.nan, -np.inf
    st = backend.image_compare(bad, ref)
    assert st.inf_count == 2 and st.nan_count == 1 and st.n == img.size
    ok = np.isfinite(bad)
    assert abs(st.mean - bad[ok].astype(np.float64).sum() / img.size) < 1e-5   # the average still runs over the whole region (:320)
    zero_ref = np.zeros_like(ref)
    assert backend.image_compare(img, zero_ref).mape == 0.0               # B == 0 pixels add nothing to MAPE (:333)


def test_global_statistics_merge_the_channels():
    sts = [backend.image_compare(*planes(seed=s)) for s in (1, 2, 3)]
    g = backend.image_stats_merge(sts)
    assert g.n == sum(s.n for s in sts)
    assert g.max == max(s.max for s in sts) and g.min_ref == min(s.min_ref for s in sts)
    assert abs(g.mse - np.mean([s.mse for s in sts])) < 1e-7 and abs(g.mape - np.mean([s.mape for s in sts])) < 1e-7


def test_the_tool_end_to_end_on_library_written_exr(tmp_path):
    img, ref = planes(shape=(24, 32))
    a, b = os.path.join(tmp_path, "a.exr"), os.path.join(tmp_path, "b.exr")
    backend.write_exr(a, {"R": img, "G": img * 0.5, "B": ref, "depth": ref + 1})
    backend.write_exr(b, {"R": ref, "G": ref * 0.5, "B": ref, "extra": ref})
    out = io.StringIO()
    with redirect_stdout(out):
        assert imgcmp.main([a, b]) == 0
    text = out.getvalue()
    assert "Channel R>" in text and "Channel G>" in text and "Channel B>" in text and "Global>" in text
    assert "Channel depth>" not in text and "Channel extra>" not in text          # only the common channels
    mse_r = backend.image_compare(img, ref).mse
    assert ("-[MSE         ] = %g" % mse_r) in text
    assert "-[MSE         ] = 0\n" in text                                       # channel B equals its reference
    out = io.StringIO()
    with redirect_stdout(out):
        assert imgcmp.main([a, b, "--channel", "G", "--ncrop", "0.25,0.25,0.75,0.75"]) == 0
    assert out.getvalue().count("Channel ") == 1 and "Global>" not in out.getvalue()
    c = os.path.join(tmp_path, "c.exr")
    backend.write_exr(c, {"R": np.zeros((5, 5), np.float32)})
    assert imgcmp.main([a, c]) == 1                                               # shapes differ
