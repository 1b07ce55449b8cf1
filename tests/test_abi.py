"""The C-ABI library loads on a CPU-only machine, exports every symbol include/prgpu.h declares and reports
errors through codes + prgpu_last_error (no compute calls without a GPU)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from pearray_amd import _cabi as abi
from pearray_amd import scene, tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = abi.load()
    header = open(os.path.join(ROOT, "include", "prgpu.h")).read()
    declared = set(re.findall(r"\b(prgpu_[a-z0-9_]+)\s*\(", header))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_the_pearray_side_adapter_and_the_docs_only_use_declared_entry_points():
    """integration/pr_pl_int_gpu_direct.cpp cannot be compiled here (PearRay's headers, Eigen); at least every prgpu_* call in it, and in
    INTEGRATION.md's host loops, must be an entry point (or type / constant) that include/prgpu.h declares."""
    header = open(os.path.join(ROOT, "include", "prgpu.h")).read()
    for rel in (os.path.join("integration", "pr_pl_int_gpu_direct.cpp"), "INTEGRATION.md"):
        text = open(os.path.join(ROOT, rel)).read()
        for name in set(re.findall(r"\b(prgpu_[a-z0-9_]+)\s*\(", text)):
            assert name in abi.SYMBOLS or name == "prgpu_init", (rel, name)   # prgpu_init: SURVEY's proposed name, mentioned as replaced
        for name in set(re.findall(r"\b(PRGPU_[A-Z0-9_]+)\b", text)):
            assert re.search(r"\b%s\b" % name, header) or name.startswith(("PRGPU_PP_", "PRGPU_MODE", "PRGPU_COMM_", "PRGPU_TRACE_", "PRGPU_LIBRARY", "PRGPU_DUMP", "PRGPU_GROUPS")), (rel, name)


def test_environment_knobs_are_read_in_one_place():
    """Every PRGPU_* environment knob of the library is read by read_knobs() in prgpu_api.hip (one table, documented there); no other
    translation unit calls getenv, and the Python side only adds PRGPU_LIBRARY (an A/B build of the same ABI)."""
    csrc = os.path.join(ROOT, "pearray_amd", "csrc")
    api = open(os.path.join(csrc, "prgpu_api.hip")).read()
    body = api[api.index("Knobs read_knobs()"):]
    body = body[:body.index("\n}\n") + 3]
    assert api.count("getenv(") == body.count("getenv("), "a getenv outside read_knobs()"
    for dirpath, _, files in os.walk(csrc):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h", ".inl")) and f != "prgpu_api.hip":
                assert "getenv" not in open(os.path.join(dirpath, f), errors="replace").read(), f
    knobs = set(re.findall(r'"(PRGPU_[A-Z0-9_]+)"', body))
    assert 15 <= len(knobs) <= 30, sorted(knobs)
    decl = api[api.index("struct Knobs {"):api.index("Knobs read_knobs()")]
    # a comment may list a family as "PRGPU_PP_SLOTS, _SHADE_MIN, _REFILL": expand the short forms to full names
    for m in re.finditer(r"PRGPU_([A-Z]+)_[A-Z0-9_]+(?:, _[A-Z0-9_]+)+", decl):
        decl += " " + " ".join("PRGPU_%s%s" % (m.group(1), short) for short in re.findall(r", (_[A-Z0-9_]+)", m.group(0)))
    for name in knobs:   # each knob is explained where it is declared
        assert name in decl, name


FOLDED_PROBES = {"gpu_probe_modes.py", "gpu_probe_shares.py", "gpu_probe_share8.py", "gpu_share_ranks.py", "gpu_sweep_share8.sh"}


def test_measurement_probes_are_the_ones_the_readme_names():
    """tools/gpu_* are reproducible entry points, not a scrap heap: each one is named in README.md, and every probe or profile file that
    README.md, DESIGN.md or INTEGRATION.md point to exists."""
    readme = open(os.path.join(ROOT, "README.md")).read()
    for f in sorted(os.listdir(os.path.join(ROOT, "tools"))):
        if f.startswith("gpu_"):
            assert ("tools/" + f) in readme or f in readme, f
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for rel in set(re.findall(r"`((?:tools|profiles|integration|oracle/ref)/[A-Za-z0-9_./-]+\.(?:py|sh|json|log|csv|txt|c|cpp))`", text)):
            if "r01_" in rel or "r02_" in rel or "{" in rel or "*" in rel or os.path.basename(rel) in FOLDED_PROBES:
                continue   # (history: DESIGN section 7 says which round-1 / round-2 probes were folded into the current ones)
            assert os.path.exists(os.path.join(ROOT, rel)), (doc, rel)


def test_a_one_rank_communicator_needs_no_rccl_and_says_so():
    """prgpu_comm_create(n_ranks = 1) makes no RCCL call (and needs no device); prgpu_comm_query then reports that no collective library is
    involved (0 ranks, rank -1) -- on a multi-GPU job it returns ncclCommCount / ncclCommUserRank, which bench.py prints."""
    lib = abi.load()
    os.environ.pop("PRGPU_COMM_FORCE_RCCL", None)
    h = C.c_void_p()
    assert lib.prgpu_comm_create(None, 1, 0, 0, C.byref(h)) == 0
    n, r = C.c_int(7), C.c_int(7)
    assert lib.prgpu_comm_query(h, C.byref(n), C.byref(r)) == 0 and (n.value, r.value) == (0, -1)
    assert lib.prgpu_comm_size(h) == 1
    lib.prgpu_comm_destroy(h)
    assert lib.prgpu_comm_query(None, C.byref(n), C.byref(r)) == -1
    assert lib.prgpu_comm_create(None, 2, 0, 0, C.byref(h)) == -1 and b"unique id" in lib.prgpu_last_error()   # several ranks need rank 0's id
    assert lib.prgpu_reduced_planes(None, None, None, None) == -1 and lib.prgpu_pipeline_info_get(None, None) == -1


def test_profiles_index_is_current():
    """profiles/INDEX.md lists every committed measurement with the documents and sources that cite it (tools/make_profiles_index.py writes it)."""
    import subprocess
    names = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if f != "INDEX.md"]
    index = open(os.path.join(ROOT, "profiles", "INDEX.md")).read()
    for f in names:
        assert "`%s`" % f in index, "profiles/%s is not in profiles/INDEX.md: run python tools/make_profiles_index.py" % f
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_profiles_index.py"), "--check"]).returncode == 0, "profiles/INDEX.md is stale: run python tools/make_profiles_index.py"


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof of every ABI struct as seen by a C compiler equal the ctypes mirror."""
    import subprocess
    structs = {"prgpu_spectrum": abi.Spectrum, "prgpu_material": abi.Material, "prgpu_emission": abi.Emission,
               "prgpu_entity": abi.Entity, "prgpu_camera": abi.Camera, "prgpu_settings": abi.Settings,
               "prgpu_scene_desc": abi.SceneDesc, "prgpu_tile": abi.Tile, "prgpu_trace_counters": abi.TraceCounters,
               "prgpu_light": abi.Light, "prgpu_image_stats": abi.ImageStats, "prgpu_output_channel": abi.OutputChannel}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "prgpu.h"', "int main(void){"]
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(cls, f).offset, (cname, f)


def test_default_settings_are_the_reference_defaults():
    lib = abi.load()
    s = abi.Settings()
    lib.prgpu_settings_default(C.byref(s))
    d = abi.default_settings(1920, 1080)
    for f, _ in abi.Settings._fields_:
        assert getattr(s, f) == getattr(d, f), f
    assert (s.seed, s.aa_sampler, s.aa_samples, s.filter, s.filter_radius) == (42, abi.SAMPLER_SOBOL, 128, abi.FILTER_MITCHELL, 1)
    assert (s.max_ray_depth, s.soft_max_ray_depth, s.mis, s.nee) == (64, 4, abi.MIS_BALANCE, 1)


def test_error_reporting_without_a_gpu():
    lib = abi.load()
    h = C.c_void_p()
    assert lib.prgpu_scene_create(None, 0, C.byref(h)) == -1
    assert b"null" in lib.prgpu_last_error()
    sc = scene.cornell_box(8, 8, spp=1)
    sc.desc.api_version = 99
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"api_version" in lib.prgpu_last_error()
    sc = scene.cornell_box(8, 8, spp=1)
    sc.indices[5] = 99999
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"index" in lib.prgpu_last_error()
    sc = scene.cornell_box(8, 8, spp=1)
    sc.desc.settings.filter_radius = 9
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -4  # PRGPU_EUNSUPPORTED
    sc = scene.cornell_box(8, 8, spp=1)
    sc.materials[0].kind = 9
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -4
    sc = scene.cornell_box(8, 8, spp=1)
    sc.positions[7] = np.inf  # would send the device BVH build (Morton codes, quantised node grid) astray
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"finite" in lib.prgpu_last_error()
    sc = scene.cornell_box(8, 8, spp=1)
    sc.desc.entities[0].transform[3] = float("nan")
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"finite" in lib.prgpu_last_error()
    sc = scene.cornell_box(8, 8, spp=1)
    sc.desc.camera.transform[7] = float("nan")  # every camera ray would start at a NaN
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"camera" in lib.prgpu_last_error()
    for near in (-1e-3, -0.0):  # primary rays start at `near`; the traversal compares entry distances by their bit pattern (>= +0 only)
        sc = scene.cornell_box(8, 8, spp=1)
        sc.desc.camera.near_t = near
        assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"near" in lib.prgpu_last_error()
    sc = scene.cornell_box(8, 8, spp=1)
    sc.desc.camera.aperture_radius = float("nan")
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"camera" in lib.prgpu_last_error()
    assert lib.prgpu_rgb_to_coeffs(None, None) == -1
    bad = (C.c_float * 3)(float("nan"), 0, 0)
    assert lib.prgpu_rgb_to_coeffs(bad, (C.c_float * 3)()) == -1
    assert lib.prgpu_render(None, 0, 1) == -1


def test_a_missing_library_is_an_import_error_not_a_fallback(tmp_path):
    """No CPU path behind the product: with the HIP library absent the host mirror refuses to load (a fresh interpreter, PRGPU_LIBRARY
    pointing at nothing)."""
    import subprocess
    code = "from pearray_amd import _cabi\ntry:\n    _cabi.load()\nexcept ImportError as e:\n    print('refused:', 'no CPU fallback' in str(e))\n"
    env = dict(os.environ, PRGPU_LIBRARY=str(tmp_path / "libprgpu_absent.so"))
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "refused: True", out.stdout + out.stderr


def test_scene_create_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = abi.load()
    h = C.c_void_p()
    rc = lib.prgpu_scene_create(C.byref(scene.cornell_box(8, 8, spp=1).desc), 0, C.byref(h))
    assert rc == -2 and not h.value and b"no CPU fallback" in lib.prgpu_last_error()


def test_communicator_argument_errors_are_clear():
    """Advisor finding: n_ranks > 1 without a way to ship rank 0's id used to end in an opaque TypeError."""
    from pearray_amd import backend
    with pytest.raises(ValueError, match="exchange"):
        backend.Communicator(2, 1)
    with pytest.raises(ValueError, match="instead of rank 0's"):
        backend.Communicator(2, 1, exchange=lambda raw: None)   # rank 0 could not create an id and sent the sentinel


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pearray_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inl")) and "build" not in dirpath:
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in src.replace("the CPU oracle", "") or f == "pr_tables.inl", os.path.join(dirpath, f)
                assert "libpr_oracle" not in src and "pr_oracle.h" not in src


def test_tiling_partitions_the_film():
    W, H = 1920, 1080
    seen = np.zeros((H, W), dtype=np.int32)
    counts = []
    for r in range(8):
        t = tiling.tiles_for_rank(W, H, r, 8)
        counts.append(tiling.owned_pixel_count(t))
        for x0, y0, x1, y1 in t:
            seen[y0:y1, x0:x1] += 1
    assert (seen == 1).all() and sum(counts) == W * H
    assert max(counts) - min(counts) <= 3 * 64 * 64
    assert tiling.tiles_for_rank(7, 5, 0, 1, tile=4) == [(0, 0, 4, 4), (4, 0, 7, 4), (0, 4, 4, 5), (4, 4, 7, 5)]
    with pytest.raises(ValueError):
        tiling.tiles_for_rank(8, 8, 2, 2)


def test_cie_tables_of_product_and_checker_are_the_same_data():
    """Both sides include their own copy of the generated CIE / D65 tables; a drift would make 'bit-exact vs oracle' compare different data."""
    a = open(os.path.join(ROOT, "oracle", "pr_tables.inl"), "rb").read()
    b = open(os.path.join(ROOT, "pearray_amd", "csrc", "tables", "pr_tables.inl"), "rb").read()
    assert a == b


def test_no_built_binaries_are_tracked():
    import subprocess
    tracked = subprocess.check_output(["git", "-C", ROOT, "ls-files"]).decode().split()
    for f in tracked:
        with open(os.path.join(ROOT, f), "rb") as fh:
            assert fh.read(4) != b"\x7fELF", f


def test_communicator_argument_checks_without_a_gpu():
    """prgpu_comm_* / prgpu_reduce: a one-rank communicator needs neither RCCL nor a device; bad arguments are EINVAL."""
    lib = abi.load()
    h = C.c_void_p()
    assert lib.prgpu_comm_create(None, 1, 0, 0, C.byref(h)) == 0 and h.value
    assert lib.prgpu_comm_size(h) == 1
    assert lib.prgpu_reduce(None, h, 0) == -1 and b"null" in lib.prgpu_last_error()
    lib.prgpu_comm_destroy(h)
    h = C.c_void_p()
    assert lib.prgpu_comm_create(None, 2, 2, 0, C.byref(h)) == -1 and b"rank" in lib.prgpu_last_error()
    assert lib.prgpu_comm_create(None, 2, 0, 0, C.byref(h)) == -1 and b"unique id" in lib.prgpu_last_error()
    assert lib.prgpu_comm_create(None, 0, 0, 0, C.byref(h)) == -1
    assert lib.prgpu_comm_size(None) == -1
    lib.prgpu_comm_destroy(None)
