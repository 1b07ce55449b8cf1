"""The evaluation scene against Mitsuba's image (examples/evaluation/cbox.exr), the coloured walls taken apart.  CPU only (the checker renders);
needs the reference checkout for the EXR.  usage: python tools/probe_cbox_walls.py [/root/reference]  > profiles/r04_cbox_walls.log
(1) first bounce analytically: sRGB of wall x light under CIE 1931 (Mitsuba) and CIE 2006 (PearRay);
(2) an RGB-mode emulation (three scalar renders with every spectrum replaced by its E-weighted sRGB channel -- what an RGB renderer computes):
    does Mitsuba's image look like it?
(3) the walls' pixel profiles, ours against Mitsuba's, front to back, at path depth limits 3 / 6 / 12."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_binding as ob
import exr_piz
from pearray_amd import scene
from pearray_amd import _cabi as abi

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
im = exr_piz.read_exr(os.path.join(ref, "examples", "evaluation", "cbox.exr"))
MI = np.stack([im["R"], im["G"], im["B"]], -1)
txt = open(os.path.join(ROOT, "pearray_amd", "csrc", "tables", "pr_tables.inl")).read()
tab = lambda n, k: np.array([float(x.strip().rstrip("f")) for x in re.search(r"%s\[%d\] = \{(.*?)\};" % (n, k), txt, re.S).group(1).split(",") if x.strip()])
X31, Y31, Z31 = (tab("PR_CIE1931_" + c, 95) for c in "XYZ"); L31 = np.arange(360, 831, 5.0)
X06, Y06, Z06 = (tab("PR_CIE2006_" + c, 441) for c in "XYZ"); L06 = np.arange(390, 831, 1.0)
M = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]])
data = json.load(open(os.path.join(ROOT, "pearray_amd", "data", "cbox_eval.json")))
lum = lambda c: c @ np.array([0.2126, 0.7152, 0.0722])


def spec(m, lam, zero_outside):
    v = np.asarray(m["values"], float)
    out = np.interp((lam - m["start"]) / (m["end"] - m["start"]) * (len(v) - 1), np.arange(len(v)), v)
    return np.where((lam < m["start"]) | (lam > m["end"]), 0.0, out) if zero_outside else out


def srgb31(s):
    return M @ (np.array([(X31 * s).sum(), (Y31 * s).sum(), (Z31 * s).sum()]) / Y31.sum())


print("(1) first bounce, sRGB of reflectance x light")
for n in ("white", "red", "green"):
    a = srgb31(spec(data["materials"][n], L31, True) * spec(data["emission"], L31, True))
    s = spec(data["materials"][n], L06, False) * spec(data["emission"], L06, False)
    b = M @ (np.array([(X06 * s).sum(), (Y06 * s).sum(), (Z06 * s).sum()]) / Y06.sum())
    p = srgb31(spec(data["materials"][n], L31, True)) * srgb31(spec(data["emission"], L31, True))
    print("  %-6s CIE 1931, zero outside 400..700 %s | CIE 2006, constant beyond the ends %s (luminance x %.3f) | product of the two sRGBs %s (luminance of the spectral one x %.3f)"
          % (n, np.round(a, 3), np.round(b, 3), lum(b) / lum(a), np.round(p, 3), lum(a) / lum(p)))


def render(materials, emission, w, spp, depth=6):
    b = scene.SceneBuilder(w, w)
    s = b.settings
    s.aa_sampler, s.aa_samples, s.max_ray_depth, s.mapper, s.filter, s.filter_radius = abi.SAMPLER_SOBOL, spp, depth, abi.MAPPER_RANDOM, abi.FILTER_TRIANGLE, 0
    mats = {n: b.lambert(f(b)) for n, f in materials.items()}
    ems = b.diffuse_emission(emission(b))
    for ent in data["entities"]:
        T = np.eye(4, dtype=np.float32)
        if ent["position"]:
            T[:3, 3] = ent["position"]
        b.add_mesh(ent["p"], ent["faces"], mats[ent["material"]], normals=ent.get("n"), emission=ems if ent["emission"] else None, transform=T)
    cam = data["camera"]
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = cam["position"]
    b.set_camera(T, width=cam["width"][0], height=cam["height"][0], near=cam["near"][0], far=cam["far"][0], local_direction=cam["local_direction"],
                 local_right=cam["local_right"], local_up=cam["local_up"])
    o = ob.OracleScene(b.build())
    o.render(spp, threads=os.cpu_count())
    return np.minimum(o.output()[0].reshape(w, w, 3), 4.0)


print("(2) RGB-mode emulation (every spectrum -> its E-weighted sRGB channel, three scalar renders) against Mitsuba's block means")
blocks = MI.reshape(16, 16, 16, 16, 3).mean(axis=(1, 3))
chan = []
for c in range(3):
    mats = {n: (lambda b, v=float(np.clip(srgb31(spec(m, L31, True))[c], 0, 1)): b.spectrum_const(v)) for n, m in data["materials"].items()}
    y = render(mats, lambda b, v=float(max(srgb31(spec(data["emission"], L31, True))[c], 0)): b.spectrum_const(v), 96, 128)[..., 1]
    chan.append(y.reshape(16, 6, 16, 6).mean(axis=(1, 3)))
emu = np.stack(chan, -1)
white = (slice(5, 12), slice(5, 8))
print("  white surfaces (blocks rows 5..11, cols 5..7): emulated RGB / Mitsuba: R %.3f G %.3f B %.3f; R : G of Mitsuba %.3f, of the emulation %.3f"
      % (tuple((emu[white][..., k] / blocks[white][..., k]).mean() for k in range(3))
         + ((blocks[white][..., 1] / blocks[white][..., 0]).mean(), (emu[white][..., 1] / emu[white][..., 0]).mean())))

print("(3) wall profiles, rows 96..159, 4-pixel columns: ours / Mitsuba in the wall's own channel, at depth limits 3 / 6 / 12")
tables = {n: (lambda b, m=m: b.spectrum_table(m["start"], m["end"], m["values"])) for n, m in data["materials"].items()}
e = data["emission"]
res = {d: render(tables, lambda b: b.spectrum_table(e["start"], e["end"], e["values"]), 256, 64, d) @ M.T for d in (3, 6, 12)}
rows = slice(96, 160)
for name, cols, k in (("red wall, front -> back  (R)", (6, 10, 16, 24, 32, 40, 48), 0), ("green wall, back -> front (G)", (215, 223, 231, 239, 245, 249), 1)):
    for d in (3, 6, 12):
        print("  %-30s depth %2d: %s" % (name, d, " ".join("%.2f" % (res[d][rows, c:c + 4, k].mean() / MI[rows, c:c + 4, k].mean()) for c in cols)))
yo = (res[6] @ np.linalg.inv(M).T)[..., 1]
ym = lum(MI)
print("  white surfaces at depth 6 (16 x 16-pixel blocks rows 5..11, cols 5..7): ours / Mitsuba luminance %.3f"
      % (np.minimum(yo, 2.0).reshape(16, 16, 16, 16).mean(axis=(1, 3))[white] / np.minimum(ym, 2.0).reshape(16, 16, 16, 16).mean(axis=(1, 3))[white]).mean())
