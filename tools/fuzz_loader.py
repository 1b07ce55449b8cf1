#!/usr/bin/env python3
"""Fuzz the host-side readers of foreign files under AddressSanitizer + UndefinedBehaviorSanitizer.

The library parses files it did not write: the .prc scene language (host/datalisp.cpp, host/prc_loader.cpp) and, through `(embed ...)`,
Wavefront OBJ, PLY (ascii / binary, both byte orders) and Mitsuba-serialized (zlib) meshes -- the counterparts of the reference's
src/loader/SceneLoader.cpp:775-846 and src/loader/archives/{WavefrontLoader,PlyLoader,MtsSerializedLoader}.cpp.  `make -C pearray_amd/csrc san`
builds exactly those translation units (plain C++, no device code) with -fsanitize=address,undefined -fno-sanitize-recover=all and
san/san_driver.cpp as their host; this script writes seed files of every format (its own, generated here), derives truncations and byte /
token mutations of them with a FIXED seed, and feeds the lot to build_san/prc_san.  A finding = the driver dies (the sanitizers abort);
a clean run = one status line per input, every one of them `0` (loaded) or a negative error code.

  python tools/fuzz_loader.py [--n 1500] [--seed 5] [--keep DIR] [--extra file.prc ...]     -> prints a summary, exit code 1 on a finding
tests/test_fuzz_loader.py runs a bounded version (a few hundred inputs) in the CPU suite."""
import argparse
import os
import random
import struct
import subprocess
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pearray_amd", "csrc")
DRIVER = os.path.join(CSRC, "build_san", "prc_san")

SCENE = """; seed scene of tools/fuzz_loader.py
(scene :name 'fuzz' :render_width 16 :render_height 12 :camera 'cam' :spectral_domain [400, 700] :spectral_hero true
  (sampler :slot 'aa' :type 'sobol' :sample_count 4)
  (filter :slot 'pixel' :type 'mitchell' :radius 1)
  (spectral_mapper :type 'spd')
  (integrator :type 'direct' :max_ray_depth 8 :light_sample_count 1)
  (output :name 'image' (channel :type 'color' :color 'xyz') (channel :type 'color' :lpe 'C<TS>*DL') (channel :type 'n') (channel :type 'feedback'))
  (camera :name 'cam' :type 'standard' :width 0.72 :height 0.54 :near 0.01 :far 100 :transform [1,0,0,0, 0,0,1,-3, 0,1,0,1, 0,0,0,1])
  (emission :name 'lamp' :type 'standard' :radiance (smul (illuminant "D65") (illum 17 12 4)))
  (material :name 'white' :type 'diffuse' :albedo (refl 0.725 0.71 0.68))
  (material :name 'glass' :type 'glass' :index (lookup_index "bk7") :specularity 1)
  (material :name 'metal' :type 'roughconductor' :roughness 0.2 :eta 0.2 :k 3.9)
  (material :name 'table' :type 'diffuse' :albedo (spectrum :start 400 :end 700 0.1 0.2 0.4 0.8 0.6 0.3))
  (mesh :name 'quad' (attribute :type 'p' [-1,-1,0],[1,-1,0],[1,1,0],[-1,1,0]) (attribute :type 'n' [0,0,1],[0,0,1],[0,0,1],[0,0,1])
        (attribute :type 't' [0,0],[1,0],[1,1],[0,1]) (faces [0,1,2],[0,2,3]) (materials 0, 0))
  (embed :loader 'obj' :file 'seed.obj' :name 'obj')
  (embed :loader 'ply' :file 'seed_a.ply' :name 'plya')
  (embed :loader 'ply' :file 'seed_l.ply' :name 'plyl')
  (embed :loader 'ply' :file 'seed_b.ply' :name 'plyb')
  (embed :loader 'mts' :file 'seed.serialized' :name 'mts' :shape 1)
  (include 'seed_inc.prc')
  (entity :name 'floor' :type 'mesh' :mesh 'quad' :materials 'white')
  (entity :name 'e1' :type 'mesh' :mesh 'obj' :materials 'glass' :position [0, 0, 0.5] :scale 0.3)
  (entity :name 'e2' :type 'mesh' :mesh 'plya' :materials 'metal' :transform [0.2,0,0,0.5, 0,0.2,0,0, 0,0,0.2,0.3, 0,0,0,1])
  (entity :name 'e3' :type 'mesh' :mesh 'plyl' :materials 'table')
  (entity :name 'e4' :type 'mesh' :mesh 'plyb' :materials 'white')
  (entity :name 'e5' :type 'mesh' :mesh 'mts' :materials 'white')
  (entity :name 'lamp' :type 'mesh' :mesh 'quad' :materials 'white' :emission 'lamp' :position [0, 0, 1.9] :scale 0.2)
  (entity :name 'ball' :type 'sphere' :radius 0.2 :position [0.4, 0.2, 0.2] :material 'glass')
  (light :name 'sky' :type 'sky' :turbidity 3 :azimuth_resolution 16 :elevation_resolution 8)
  (light :name 'sun' :type 'sun' :radius 1)
  (light :name 'env' :type 'env' :radiance (illuminant "D65"))
)
"""
INCLUDE = "(material :name 'inc' :type 'mirror' :specularity 0.9)\n(entity :name 'e6' :type 'plane' :width 2 :height 2 :position [0,1,1] :material 'inc')\n"
OBJ = """# seed mesh
o one
v -1 0 -1
v 1 0 -1
v 1 0 1
v -1 0 1
v 0 1 0
vn 0 1 0
vt 0 0
vt 1 0
vt 1 1
usemtl a
f 1/1/1 2/2/1 3/3/1 4
f 1//1 2//1 5//1
g two
f -1 -2 -3
l 1 2
"""


def ply(fmt):
    P = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 0.5, 1]]
    N = [[0, 0, 2], [0, 0, 1], [0, 3, 0], [0, 0, 0], [1, 1, 1]]
    UV = [[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]]
    F = [[0, 1, 2, 3], [0, 1, 4], [1, 2, 4]]
    header = ("ply\nformat %s 1.0\ncomment seed\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nproperty float confidence\n"
              "property float nx\nproperty float ny\nproperty float nz\nproperty float u\nproperty float v\nelement face 3\n"
              "property list uchar int vertex_indices\nend_header\n" % fmt).encode()
    if fmt == "ascii":
        body = "".join("%g %g %g 0.5 %g %g %g %g %g\n" % (*p, *n, *uv) for p, n, uv in zip(P, N, UV)) + "".join("%d %s\n" % (len(f), " ".join(map(str, f))) for f in F)
        return header + body.encode()
    e = "<" if "little" in fmt else ">"
    return header + b"".join(struct.pack(e + "9f", *p, 0.5, *n, *uv) for p, n, uv in zip(P, N, UV)) + b"".join(struct.pack(e + "B%di" % len(f), len(f), *f) for f in F)


def serialized(version=4, double=False, with_normals=True):
    shapes = [([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]]),
              ([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0], [1, 1, 3]], [[0, 1, 2], [0, 2, 3], [0, 1, 4], [2, 3, 4]])]
    recs, blob = [], b""
    for k, (P, F) in enumerate(shapes):
        recs.append(len(blob))
        flags = (0x2000 if double else 0x1000) | (0x0001 if with_normals else 0) | 0x0008
        raw = struct.pack("<I", flags) + (b"shape%d\0" % k if version >= 4 else b"") + struct.pack("<QQ", len(P), len(F))
        ft = "d" if double else "f"
        raw += struct.pack("<%d%s" % (3 * len(P), ft), *[float(c) for p in P for c in p])
        if with_normals:
            raw += struct.pack("<%d%s" % (3 * len(P), ft), *([0.0, 0.0, 1.0] * len(P)))
        raw += struct.pack("<%d%s" % (3 * len(P), ft), *([0.5] * 3 * len(P)))
        raw += struct.pack("<%dI" % (3 * len(F)), *[i for f in F for i in f])
        blob += struct.pack("<HH", 0x041C, version) + zlib.compress(raw)
    return blob + b"".join(struct.pack("<Q" if version >= 4 else "<I", r) for r in recs) + struct.pack("<I", len(shapes))


SEEDS = {"seed.obj": OBJ.encode(), "seed_a.ply": ply("ascii"), "seed_l.ply": ply("binary_little_endian"), "seed_b.ply": ply("binary_big_endian"),
         "seed.serialized": serialized(), "seed_inc.prc": INCLUDE.encode()}
TOKENS = [b"(", b")", b"[", b"]", b"'", b'"', b":", b",", b";", b"-", b"1e39", b"nan", b"inf", b"-1", b"0", b"4294967296", b"99999999999999999999", b"(embed", b"(include 'scene.prc')",
          b":shape 9", b":radius -1", b"(refl", b"(smul", b"\x00", b"\xff", b"\n", b"true", b"(spectrum :start 700 :end 400 1 2)", b"(mesh", b"(faces [0,1,99])", b":lpe 'C((('"]


def mutate(data, rng):
    """One mutant of a byte string: truncation, byte flips, an inserted / duplicated / deleted span, a 32-bit field set to an edge value."""
    b = bytearray(data)
    kind = rng.randrange(7)
    if kind == 0 and len(b) > 1:
        return bytes(b[:rng.randrange(len(b))])
    if kind == 1:
        for _ in range(1 + rng.randrange(4)):
            if b:
                b[rng.randrange(len(b))] = rng.randrange(256)
        return bytes(b)
    if kind == 2:
        pos = rng.randrange(len(b) + 1)
        return bytes(b[:pos] + rng.choice(TOKENS) + b[pos:])
    if kind == 3 and len(b) > 4:
        i = rng.randrange(len(b) - 1)
        j = min(len(b), i + 1 + rng.randrange(16))
        return bytes(b[:i] + b[j:])
    if kind == 4 and len(b) > 4:
        i = rng.randrange(len(b) - 1)
        j = min(len(b), i + 1 + rng.randrange(32))
        return bytes(b[:j] + b[i:j] * (1 + rng.randrange(3)) + b[j:])
    if kind == 5 and len(b) >= 4:
        i = rng.randrange(len(b) - 3)
        b[i:i + 4] = struct.pack("<I", rng.choice([0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE, 65536, 0x01000000]))
        return bytes(b)
    if b:                                   # swap two spans
        i, j = sorted((rng.randrange(len(b)), rng.randrange(len(b))))
        return bytes(b[:i] + b[j:] + b[i:j])
    return bytes(b)


def make_case(directory, k, rng):
    """Case k: the seed scene with ONE of its files (or the scene text itself) replaced by a mutant; returns the .prc path."""
    d = os.path.join(directory, "c%05d" % k)
    os.makedirs(d)
    files = dict(SEEDS)
    files["scene.prc"] = SCENE.encode()
    victim = rng.choice(["scene.prc"] * 4 + sorted(SEEDS))        # the scene text most often: it reaches every block parser
    if victim == "seed.serialized" and rng.random() < 0.5:        # mutate inside the zlib stream as well as around it
        files[victim] = serialized(version=rng.choice([3, 4]), double=rng.random() < 0.5, with_normals=rng.random() < 0.5)
    m = files[victim]
    for _ in range(1 + rng.randrange(3)):
        m = mutate(m, rng)
    files[victim] = m
    for name, data in files.items():
        with open(os.path.join(d, name), "wb") as f:
            f.write(data)
    return os.path.join(d, "scene.prc")


def run(n=1500, seed=5, keep=None, extra=(), timeout=900):
    if not os.path.exists(DRIVER):
        subprocess.check_call(["make", "-C", CSRC, "san"], stdout=subprocess.DEVNULL)
    rng = random.Random(seed)
    tmp = keep or tempfile.mkdtemp(prefix="prc_fuzz_")
    os.makedirs(tmp, exist_ok=True)
    d0 = os.path.join(tmp, "seed")
    os.makedirs(d0, exist_ok=True)
    for name, data in list(SEEDS.items()) + [("scene.prc", SCENE.encode())]:
        with open(os.path.join(d0, name), "wb") as f:
            f.write(data)
    paths = [os.path.join(d0, "scene.prc")] + [make_case(tmp, k, rng) for k in range(n)] + list(extra)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=2048", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([DRIVER, "-"], input=("\n".join(paths) + "\n").encode(), capture_output=True, env=env, timeout=timeout)
    lines = [l for l in p.stdout.decode("ascii", "replace").split("\n") if l]
    codes = {}
    for l in lines:
        codes[l.split()[0]] = codes.get(l.split()[0], 0) + 1
    finding = p.returncode != 0 or len(lines) != len(paths)
    report = {"inputs": len(paths), "answered": len(lines), "status_codes": codes, "driver_exit": p.returncode, "seed": seed,
              "first_unanswered": paths[len(lines)] if len(lines) < len(paths) else None, "seed_scene": lines[0].split(" | ")[0].split()[0] + (" | " + lines[0].split(" | ", 1)[1] if " | " in lines[0] else "") if lines else None,
              "stderr_tail": p.stderr.decode("utf-8", "replace")[-3000:] if finding else ""}
    if not keep and not finding:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return finding, report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--extra", nargs="*", default=[])
    a = ap.parse_args()
    bad, rep = run(a.n, a.seed, a.keep, a.extra)
    print(rep)
    sys.exit(1 if bad else 0)
