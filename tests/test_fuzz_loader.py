"""The host-side readers of foreign files (.prc scene language, OBJ / PLY / Mitsuba-serialized archives) under AddressSanitizer +
UndefinedBehaviorSanitizer: `make -C pearray_amd/csrc san` builds them (plain C++, no device) with san/san_driver.cpp as their host, and
tools/fuzz_loader.py feeds the driver truncations and byte / token mutations of seed files it generates itself, with a fixed seed.
A finding aborts the driver; a clean run answers every input with 0 or a negative error code.  (Counterparts in the reference:
src/loader/SceneLoader.cpp:775-846, src/loader/archives/{WavefrontLoader,PlyLoader,MtsSerializedLoader}.cpp.)  The full-size run
(`python tools/fuzz_loader.py --n 3000`) is logged in profiles/r05_fuzz_loader.log."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_loader", os.path.join(ROOT, "tools", "fuzz_loader.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_seed_files_load_and_mutants_never_trip_a_sanitizer():
    m = _tool()
    finding, rep = m.run(n=400, seed=5)
    assert not finding, rep
    assert rep["seed_scene"].startswith("0"), rep            # the unmutated seed scene (every embed format, include, sky / sun / env lights) loads
    assert rep["answered"] == rep["inputs"] == 401
    assert set(rep["status_codes"]) <= {"0", "-1", "-4", "-5"}, rep   # loaded, invalid, unsupported, i/o -- never anything else
    assert rep["status_codes"].get("0", 0) >= 10 and rep["status_codes"].get("-1", 0) >= 100, rep   # mutants reach both the success path and the refusals


def test_the_sanitizer_build_would_report_a_finding():
    """The harness itself: a driver that dies must be reported as a finding (a missing status line), not as a clean run."""
    m = _tool()
    real = m.DRIVER
    try:
        m.DRIVER = "/bin/false"
        finding, rep = m.run(n=3, seed=1)
    finally:
        m.DRIVER = real
    assert finding and rep["answered"] == 0
