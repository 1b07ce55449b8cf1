"""Host-side scene assembly: flattens meshes / materials / spectra into ``prgpu_scene_desc``.

Mirrors what PearRay's loader hands to ``Scene`` and ``RenderContext`` for the `direct` path
(src/loader/SceneLoader.cpp:446-739 entity/material/emission/node creation), restricted to what the
hot path evaluates.  The description is plain data; the same object can be given to the HIP backend
(``pearray_amd.backend``) and, in tests, to the CPU checker.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _cabi as abi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
IDENTITY = np.eye(4, dtype=np.float32)


def _d65_table():
    with open(os.path.join(_DATA, "illuminants.json")) as f:
        return json.load(f)["D65"]


def rgb_to_coeffs(rgb):
    """`(refl r g b)` coefficient lookup (SpectralValueNode.cpp:16-30) through the backend library."""
    lib = abi.load()
    src = (C.c_float * 3)(*[float(x) for x in rgb])
    dst = (C.c_float * 3)()
    abi.check(lib.prgpu_rgb_to_coeffs(src, dst))
    return [dst[0], dst[1], dst[2]]


class SceneBuilder:
    def __init__(self, width, height):
        self.settings = abi.default_settings(width, height)
        self.camera = abi.Camera()
        self.spectra, self.tables = [], []
        self.materials, self.emissions, self.entities, self.lights = [], [], [], []
        self._pos, self._nrm, self._idx, self._trimat = [], [], [], []
        self._n_vertices = 0
        self._any_normals = False
        self._uv, self._any_uvs = [], False
        self.set_camera(IDENTITY)

    # ---- spectral nodes -------------------------------------------------------------------------
    def _add_spec(self, **kw):
        s = abi.Spectrum()
        for k, v in kw.items():
            if k == "p":
                for i, x in enumerate(v):
                    s.p[i] = x
            else:
                setattr(s, k, v)
        self.spectra.append(s)
        return len(self.spectra) - 1

    def spectrum_const(self, value):
        return self._add_spec(kind=abi.SPEC_CONST, p=[value])

    def spectrum_coeffs(self, coeffs, power=None):
        if power is None:
            return self._add_spec(kind=abi.SPEC_PARAMETRIC, p=list(coeffs))
        return self._add_spec(kind=abi.SPEC_PARAMETRIC_SCALED, p=list(coeffs) + [power])

    def refl(self, r, g, b):
        """(refl r g b), SpectralValueNode.cpp:16-30"""
        return self.spectrum_coeffs(rgb_to_coeffs((r, g, b)))

    def illum(self, r, g, b):
        """(illum r g b), SpectralValueNode.cpp:31-47: scaled so the fitted colour has max 0.5"""
        rgb = np.array([r, g, b], dtype=np.float32)
        mx = np.float32(rgb.max())
        if mx <= 0:
            return self.spectrum_coeffs(rgb_to_coeffs(rgb), power=1.0)
        scale = np.float32(2) * mx
        return self.spectrum_coeffs(rgb_to_coeffs(rgb / scale), power=float(scale))

    def spectrum_table(self, start, end, values):
        """(spectrum :start :end v...), SpectralConstNode.cpp:12-33 (values clamped to >= 0)"""
        off = len(self.tables)
        self.tables.extend(max(0.0, float(v)) for v in values)
        return self._add_spec(kind=abi.SPEC_TABLE, table_offset=off, table_count=len(values), wl_start=start, wl_end=end)

    def illuminant_d65(self):
        """(illuminant "D65"), IlluminantNode.cpp:59,89-90: 107 samples 300..830 nm"""
        return self.spectrum_table(300.0, 830.0, _d65_table())

    def checkerboard(self, op1, op2, scale_u=None, scale_v=None):
        """(checkerboard a b [su [sv]]), CheckerboardNode.cpp:78-90: 2 arguments = unscaled uv, 3 = isotropic, 4 = anisotropic"""
        mode = 0 if scale_u is None else (1 if scale_v is None else 2)
        su = 5.0 if scale_u is None else float(scale_u)
        sv = su if scale_v is None else float(scale_v)
        return self._add_spec(kind=abi.SPEC_CHECKER, lhs=op1, rhs=op2, p=[su, sv, float(mode)])

    def smul(self, a, b):
        return self._add_spec(kind=abi.SPEC_MUL, lhs=a, rhs=b)

    # ---- materials / emissions ---------------------------------------------------------------------
    def lambert(self, albedo, two_sided=True):
        m = abi.Material(abi.MAT_LAMBERT, albedo, 1 if two_sided else 0, 0, 0, 0, 0)
        self.materials.append(m)
        return len(self.materials) - 1

    SELLMEIER = {  # (lookup_index name): B and C coefficients, ReflectiveNode.cpp:224-232 (refractiveindex.info data)
        "bk7": ([1.03961212, 0.231792344, 1.01046945], [0.00600069867, 0.0200179144, 103.560653]),
        "h2o": ([5.684027565e-1, 1.726177391e-1, 2.086189578e-2, 1.130748688e-1], [5.101829712e-3, 1.821153936e-2, 2.620722293e-2, 1.069792721e1]),
        "diamond": ([0.3306, 4.3356], [0.030625, 0.011236]),
    }

    def sellmeier(self, bs, cs):
        """(sellmeier_index ...) / (lookup_index name): n = sqrt(1 + sum B l^2 / (l^2 - C)), ReflectiveNode.cpp:105-150"""
        off = len(self.tables)
        self.tables.extend(float(v) for v in list(bs) + list(cs))
        return self._add_spec(kind=abi.SPEC_SELLMEIER, table_offset=off, table_count=2 * len(bs))

    def lookup_index(self, name):
        name = {"glass": "bk7", "water": "h2o"}.get(name.lower(), name.lower())
        if name in ("vacuum", "none"):
            return self.spectrum_const(1.0)
        if name == "air":
            return self.spectrum_const(1.000277)
        return self.sellmeier(*self.SELLMEIER[name])

    def dielectric(self, ior, specularity=None, transmission=None, thin=False):
        """(material :type 'glass'), dielectric.cpp:150-197 (index default 1.55, tints default 1)"""
        spec = self.spectrum_const(1.0) if specularity is None else specularity
        m = abi.Material(abi.MAT_DIELECTRIC, spec, 0, ior, abi.INVALID_ID if transmission is None else transmission, 1 if thin else 0, 0)
        self.materials.append(m)
        return len(self.materials) - 1

    def conductor(self, eta=None, k=None, specularity=None):
        """(material :type 'conductor'|'metal'), conductor.cpp:95-125 (eta 1.2, k 2.605, tint 1 by default)"""
        eta = self.spectrum_const(1.2) if eta is None else eta
        k = self.spectrum_const(2.605) if k is None else k
        spec = self.spectrum_const(1.0) if specularity is None else specularity
        self.materials.append(abi.Material(abi.MAT_CONDUCTOR, spec, 0, eta, abi.INVALID_ID, 0, k))
        return len(self.materials) - 1

    def mirror(self, specularity=None):
        """(material :type 'mirror'|'reflection'), mirror.cpp:79-100"""
        spec = self.spectrum_const(1.0) if specularity is None else specularity
        self.materials.append(abi.Material(abi.MAT_MIRROR, spec, 0, 0, abi.INVALID_ID, 0, 0))
        return len(self.materials) - 1

    def _rough(self, roughness, roughness_y, vndf):
        rx = float(roughness)
        ry, flags = (rx, 0) if roughness_y is None else (float(roughness_y), abi.MATF_ANISOTROPIC)
        return rx, ry, flags | (0 if vndf else abi.MATF_NO_VNDF)

    def rough_conductor(self, roughness, eta=None, k=None, specularity=None, roughness_y=None, vndf=True):
        """(material :type 'roughconductor'|'roughmirror'|'roughmetal') or a conductor with a roughness, roughconductor.cpp:158-216"""
        eta = self.spectrum_const(1.2) if eta is None else eta
        k = self.spectrum_const(2.605) if k is None else k
        spec = self.spectrum_const(1.0) if specularity is None else specularity
        rx, ry, flags = self._rough(roughness, roughness_y, vndf)
        self.materials.append(abi.Material(abi.MAT_ROUGH_CONDUCTOR, spec, 0, eta, abi.INVALID_ID, 0, k, flags, rx, ry))
        return len(self.materials) - 1

    def rough_dielectric(self, roughness, ior=None, specularity=None, transmission=None, roughness_y=None, vndf=True):
        """(material :type 'roughglass'|'roughdielectric') or a glass with a roughness, roughdielectric.cpp:282-365"""
        ior = self.spectrum_const(1.55) if ior is None else ior
        spec = self.spectrum_const(1.0) if specularity is None else specularity
        rx, ry, flags = self._rough(roughness, roughness_y, vndf)
        self.materials.append(abi.Material(abi.MAT_ROUGH_DIELECTRIC, spec, 0, ior, abi.INVALID_ID if transmission is None else transmission, 0, 0,
                                           flags, rx, ry))
        return len(self.materials) - 1

    def principled(self, base=None, ior=None, roughness=0.5, thin=False, vndf=True, **params):
        """(material :type 'principled'), principled.cpp:634-687; `params`: the scalars of _cabi.PRINCIPLED_PARAMS.  Giving either
        transmission parameter (even 0) selects the HasTransmission closure, like the reference's hasParameter test."""
        base = self.spectrum_const(0.8) if base is None else base
        ior = self.spectrum_const(1.55) if ior is None else ior
        unknown = set(params) - set(abi.PRINCIPLED_PARAMS)
        if unknown:
            raise TypeError("unknown principled parameters: %s" % sorted(unknown))
        flags = (0 if vndf else abi.MATF_NO_VNDF) | (abi.MATF_HAS_TRANSMISSION if ("diffuse_transmission" in params or "specular_transmission" in params) else 0)
        m = abi.Material(abi.MAT_PRINCIPLED, base, 0, ior, abi.INVALID_ID, 1 if thin else 0, 0, flags, float(roughness), float(roughness))
        for i, name in enumerate(abi.PRINCIPLED_PARAMS):
            m.principled[i] = float(params.get(name, 0.0))
        self.materials.append(m)
        return len(self.materials) - 1

    def diffuse_emission(self, radiance):
        self.emissions.append(abi.Emission(abi.EMS_DIFFUSE, radiance))
        return len(self.emissions) - 1

    # ---- geometry ------------------------------------------------------------------------------------
    def add_mesh(self, positions, faces, material, normals=None, emission=None, transform=IDENTITY, face_materials=None, uvs=None):
        """One `(entity :type 'mesh')`.  Quads are split like Embree quads: (v0,v1,v3) and (v2,v3,v1)."""
        positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
        tris, tmat = [], []
        for fi, f in enumerate(faces):
            m = material if face_materials is None else face_materials[fi]
            if len(f) == 3:
                tris.append(list(f)); tmat.append(m)
            elif len(f) == 4:
                tris.append([f[0], f[1], f[3]]); tmat.append(m)
                tris.append([f[2], f[3], f[1]]); tmat.append(m)
            else:
                raise ValueError("only triangles and quads")
        tris = np.asarray(tris, dtype=np.uint32).reshape(-1, 3)
        e = abi.Entity()
        e.first_tri = sum(len(i) for i in self._idx)
        e.n_tris = len(tris)
        e.emission = abi.INVALID_ID if emission is None else emission
        e.has_normals = 0 if normals is None else 1
        e.has_uvs = 0 if uvs is None else 1
        t = np.asarray(transform, dtype=np.float32).reshape(16)
        for i in range(16):
            e.transform[i] = float(t[i])
        self.entities.append(e)
        self._pos.append(positions)
        if normals is not None:
            self._any_normals = True
            self._nrm.append(np.asarray(normals, dtype=np.float32).reshape(-1, 3))
        else:
            self._nrm.append(np.zeros_like(positions))
        if uvs is not None:
            self._any_uvs = True
            self._uv.append(np.asarray(uvs, dtype=np.float32).reshape(-1, 2))
        else:
            self._uv.append(np.zeros((len(positions), 2), dtype=np.float32))
        self._idx.append(tris + np.uint32(self._n_vertices))
        self._trimat.append(np.asarray([abi.INVALID_ID if m is None else m for m in tmat], dtype=np.uint32))
        self._n_vertices += len(positions)
        return len(self.entities) - 1

    def _light(self, kind, radiance, background, direction, transform):
        l = abi.Light(kind, radiance, abi.INVALID_ID if background is None else background, 0)
        for i in range(3):
            l.direction[i] = float(direction[i])
        t = np.asarray(transform, dtype=np.float32).reshape(16)
        for i in range(16):
            l.transform[i] = float(t[i])
        self.lights.append(l)
        return len(self.lights) - 1

    def environment_light(self, radiance, background=None, transform=IDENTITY, image=None, distribution=True, compensation=False):
        """(light :type 'env' :radiance r [:background b]), environment.cpp:152-205.  `image`: float32 [height, width, 3] Jakob-Hanika
        coefficients (a latitude / longitude map in FILE scanline order under the reference's lookup t = 1 - v, v = theta / pi: row 0 = the
        nadir, -z, the LAST row = the zenith, +z; environment.cpp:53-101) that multiply `radiance` texel by texel -- the textured
        environment light; with more than one row and column it is importance-sampled unless `distribution` is False."""
        k = self._light(abi.LIGHT_ENVIRONMENT, radiance, background, (0, 0, 1), transform)
        if image is not None:
            image = np.ascontiguousarray(image, dtype=np.float32)
            assert image.ndim == 3 and image.shape[2] == 3
            l = self.lights[k]
            l.flags = abi.ENVF_TEXTURED | (0 if distribution else abi.ENVF_NO_DISTRIBUTION) | (abi.SKYF_COMPENSATION if compensation else 0)
            l.table_offset = len(self.tables)
            l.elevation_count, l.azimuth_count = image.shape[0], image.shape[1]
            self.tables.extend(image.reshape(-1).tolist())
        return k

    def rgb_image_to_coefficients(self, rgb):
        """An RGB image (float [H, W, 3], linear sRGB in [0, 1]) as the coefficient image a textured light takes (prgpu_rgb_to_coeffs per
        distinct colour: NonParametricImageNode converts every lookup the same way, ImageNode.cpp:150-175)."""
        rgb = np.asarray(rgb, dtype=np.float32)
        out = np.zeros_like(rgb)
        cache = {}
        for idx in np.ndindex(rgb.shape[:2]):
            key = tuple(float(c) for c in rgb[idx])
            if key not in cache:
                cache[key] = rgb_to_coeffs(key)
            out[idx] = cache[key]
        return out

    def distant_light(self, irradiance, direction=(0, 0, 1), transform=IDENTITY):
        """(light :type 'distant' :direction d :irradiance e), distant.cpp:112-121"""
        return self._light(abi.LIGHT_DISTANT, irradiance, None, direction, transform)

    SUN_VIS_RADIUS = np.float32(np.float32(np.pi) / np.float32(180.0)) * np.float32(0.5358) * np.float32(0.5)  # sun.cpp:24

    @staticmethod
    def ea_to_direction(elevation, azimuth):
        """ElevationAzimuth::toDirection (ElevationAzimuth.h:32-35): Spherical::cartesian(pi/2 - elevation, azimuth), up is +z"""
        theta = 0.5 * np.pi - elevation
        return (np.sin(theta) * np.cos(azimuth), np.sin(theta) * np.sin(azimuth), np.cos(theta))

    def sky_light(self, table, extend=True, compensation=False, transform=IDENTITY):
        """(light :type 'sky'), sky.cpp:180-198.  `table`: float32 [elevation_count, azimuth_count, 11] = SkyModel::mData (the
        Hosek-Wilkie evaluation stays with the host, SkyModel.cpp:17-60)."""
        table = np.ascontiguousarray(table, dtype=np.float32)
        assert table.ndim == 3 and table.shape[2] == abi.SKY_BANDS
        k = self._light(abi.LIGHT_SKY, abi.INVALID_ID, None, (0, 0, 1), transform)
        l = self.lights[k]
        l.flags = (abi.SKYF_EXTEND if extend else 0) | (abi.SKYF_COMPENSATION if compensation else 0)
        l.table_offset = len(self.tables)
        l.elevation_count, l.azimuth_count = table.shape[0], table.shape[1]
        self.tables.extend(table.reshape(-1).tolist())
        return k

    def sun_light(self, radiance_64, elevation, azimuth, radius=1.0, transform=IDENTITY):
        """(light :type 'sun' :radius r > 0), sun.cpp:26-137.  `radiance_64`: the 64 samples of 360..760 nm, i.e. computeSunRadiance *
        power_scale / radius^2 (sun.cpp:42-46; the Preetham tables stay with the host)."""
        assert len(radiance_64) == 64 and radius > 1.1920929e-7
        node = self.spectrum_table(360.0, 760.0, radiance_64)
        k = self._light(abi.LIGHT_SUN, node, None, self.ea_to_direction(elevation, azimuth), transform)
        self.lights[k].cos_theta = float(np.cos(np.float32(self.SUN_VIS_RADIUS * np.float32(radius))))
        return k

    def cie_sky_light(self, zenith, ground_tint=None, ground_brightness=0.2, cloudy=False, transform=IDENTITY):
        """(light :type 'uniform_sky' | 'cloudy_sky' :zenith z [:ground_tint g] [:ground_brightness b]), cie_sky.cpp:134-160"""
        k = self._light(abi.LIGHT_CIE_SKY, zenith, ground_tint, (0, 0, 1), transform)
        self.lights[k].flags = abi.SKYF_CLOUDY if cloudy else 0
        self.lights[k].ground_brightness = float(ground_brightness)
        return k

    def add_plane(self, material, x_axis=(1, 0, 0), y_axis=(0, 1, 0), width=1.0, height=1.0, centering=False, transform=IDENTITY, emission=None):
        """(entity :type 'plane'), plane.cpp:241-258: parallelogram spanned by width * x_axis and height * y_axis"""
        x = np.float32(width) * np.asarray(x_axis, dtype=np.float32)
        y = np.float32(height) * np.asarray(y_axis, dtype=np.float32)
        p = (np.float32(-0.5) * x - np.float32(0.5) * y) if centering else np.zeros(3, dtype=np.float32)   # PlaneEntity::centerOn
        e = self.add_mesh([p, p + y, (p + y) + x, p + x], [[0, 1, 3], [2, 3, 1]], material, transform=transform, emission=emission)   # plane.cpp:81-84
        self.entities[e].kind = abi.ENTITY_PLANE
        return e

    def add_sphere(self, material, radius=1.0, transform=IDENTITY, emission=None):
        """(entity :type 'sphere' :radius r), sphere.cpp:157-168: one placeholder triangle stands for the analytic primitive"""
        e = self.add_mesh([[0, 0, 0], [0, 0, 0], [0, 0, 0]], [[0, 1, 2]], material, transform=transform, emission=emission)
        self.entities[e].kind = abi.ENTITY_SPHERE
        self.entities[e].radius = float(radius)
        return e

    def add_quadric(self, material, parameters, box_min=(-1, -1, -1), box_max=(1, 1, 1), transform=IDENTITY):
        """(entity :type 'quadric' :parameters [A..J] :min :max), quadric.cpp:296-313: 3, 4 or 10 coefficients"""
        q = [float(v) for v in parameters]
        if len(q) == 3:
            q = q + [0.0] * 7
        elif len(q) == 4:
            q = q[:3] + [0.0] * 6 + q[3:]
        assert len(q) == 10, "3, 4 or 10 quadric parameters"
        e = self.add_mesh([[0, 0, 0], [0, 0, 0], [0, 0, 0]], [[0, 1, 2]], material, transform=transform)   # placeholder: degenerate, never hit
        self.entities[e].kind = abi.ENTITY_QUADRIC
        self.entities[e].params = len(self.tables)
        self.tables.extend(float(np.float32(v)) for v in q + list(box_min) + list(box_max))
        return e

    def add_cylinder(self, material, radius=1.0, height=1.0, center_on=True, transform=IDENTITY):
        """(entity :type 'cylinder'), quadric.cpp:255-272"""
        r, h = np.float32(radius), np.float32(height)
        a2 = np.float32(1) / (r * r)
        z0, z1 = (-h / np.float32(2), h / np.float32(2)) if center_on else (np.float32(0), h)
        return self.add_quadric(material, [a2, a2, 0, 0, 0, 0, 0, 0, 0, -1], (-r, -r, z0), (r, r, z1), transform)

    def add_cone(self, material, radius=1.0, height=1.0, center_on=True, transform=IDENTITY):
        """(entity :type 'cone'), quadric.cpp:273-294"""
        r, h = np.float32(radius), np.float32(height)
        a2, h2 = np.float32(1) / (r * r), np.float32(1) / (h * h)
        if center_on:
            return self.add_quadric(material, [a2, a2, -h2, 0, 0, 0, 0, 0, np.float32(1) / h, -0.25], (-r, -r, -h / np.float32(2)), (r, r, h / np.float32(2)), transform)
        return self.add_quadric(material, [a2, a2, -h2, 0, 0, 0, 0, 0, 0, 0], (-r, -r, 0), (r, r, h), transform)

    def set_camera(self, transform, width=1.0, height=1.0, near=1e-6, far=float("inf"), local_direction=(0, 0, 1),
                   local_right=(1, 0, 0), local_up=(0, 1, 0), fstop=0.0, aperture_radius=0.05, ortho=False):
        c = self.camera
        t = np.asarray(transform, dtype=np.float32).reshape(16)
        for i in range(16):
            c.transform[i] = float(t[i])
        c.width, c.height, c.near_t, c.far_t = width, height, near, far
        for i in range(3):
            c.local_direction[i] = local_direction[i]
            c.local_right[i] = local_right[i]
            c.local_up[i] = local_up[i]
        c.fstop, c.aperture_radius = fstop, aperture_radius
        c.kind = abi.CAMERA_ORTHO if ortho else abi.CAMERA_PERSPECTIVE

    def set_spherical_camera(self, transform, theta_start=0.0, theta_end=float(np.float32(np.pi) / np.float32(2)), phi_start=-float(np.float32(np.pi)),
                             phi_end=float(np.float32(np.pi)), **frame):
        """(camera :type 'spherical'), spherical.cpp:91-105"""
        self.set_camera(transform, **frame)
        c = self.camera
        c.kind = abi.CAMERA_SPHERICAL
        c.theta_start, c.theta_end, c.phi_start, c.phi_end = theta_start, theta_end, phi_start, phi_end

    def set_fisheye_camera(self, transform, fov=float(np.float32(180.0) * (np.float32(np.pi) / np.float32(180.0))), map_type=abi.FISHEYE_CIRCULAR, clip_range=True, **frame):
        """(camera :type 'fisheye' :fov f :map 'circular'|'cropped'|'full' :clip_range b), fisheye.cpp:137-173"""
        self.set_camera(transform, **frame)
        c = self.camera
        c.kind = abi.CAMERA_FISHEYE
        c.fov, c.fisheye_map, c.clip_range = fov, map_type, 1 if clip_range else 0

    def build(self):
        return SceneData(self)


class SceneData:
    """Owns the numpy/ctypes storage behind one ``prgpu_scene_desc``."""

    def __init__(self, b):
        self.positions = np.ascontiguousarray(np.concatenate(b._pos), dtype=np.float32)
        self.normals = np.ascontiguousarray(np.concatenate(b._nrm), dtype=np.float32) if b._any_normals else None
        self.uvs = np.ascontiguousarray(np.concatenate(b._uv), dtype=np.float32) if b._any_uvs else None
        self.indices = np.ascontiguousarray(np.concatenate(b._idx), dtype=np.uint32)
        self.tri_material = np.ascontiguousarray(np.concatenate(b._trimat), dtype=np.uint32)
        self.entities = (abi.Entity * len(b.entities))(*b.entities)
        self.materials = (abi.Material * max(1, len(b.materials)))(*b.materials)
        self.emissions = (abi.Emission * max(1, len(b.emissions)))(*b.emissions)
        self.spectra = (abi.Spectrum * max(1, len(b.spectra)))(*b.spectra)
        self.tables = np.ascontiguousarray(np.asarray(b.tables if b.tables else [0.0], dtype=np.float32))
        d = abi.SceneDesc()
        d.api_version = abi.PRGPU_API_VERSION
        d.n_vertices = len(self.positions)
        d.positions = self.positions.ctypes.data_as(C.POINTER(C.c_float))
        d.normals = self.normals.ctypes.data_as(C.POINTER(C.c_float)) if self.normals is not None else None
        d.uvs = self.uvs.ctypes.data_as(C.POINTER(C.c_float)) if self.uvs is not None else None
        d.n_triangles = len(self.indices)
        d.indices = self.indices.ctypes.data_as(C.POINTER(C.c_uint32))
        d.tri_material = self.tri_material.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_entities = len(b.entities)
        d.entities = self.entities
        d.n_materials = len(b.materials)
        d.materials = self.materials
        d.n_emissions = len(b.emissions)
        d.emissions = self.emissions
        d.n_spectra = len(b.spectra)
        d.spectra = self.spectra
        d.n_spectral_table_values = len(b.tables)
        d.spectral_tables = self.tables.ctypes.data_as(C.POINTER(C.c_float))
        d.camera = b.camera
        d.settings = b.settings
        self.lights = (abi.Light * max(1, len(b.lights)))(*b.lights)
        d.n_lights = len(b.lights)
        d.lights = self.lights
        self.desc = d

    @property
    def settings(self):
        return self.desc.settings

    @property
    def width(self):
        return self.desc.settings.width

    @property
    def height(self):
        return self.desc.settings.height

    @property
    def spp(self):
        s = self.desc.settings
        return s.aa_samples * s.lens_samples * s.time_samples * s.spectral_samples


class PrcScene:
    """A scene parsed from PearRay's .prc language by the backend library (prgpu_prc_*, csrc/host/prc_loader.cpp).
    Quacks like SceneData: .desc, .settings, .width, .height, .spp."""

    def __init__(self, path=None, source=None, include_dir=None, width=0, height=0, spp=0, force_direct=False, seed=0, skies=None):
        """skies: {light name or None: float32 [elevation, azimuth, 11]} -- the SkyModel tables of the scene's sky lights"""
        lib = abi.load()
        opt = abi.PrcOptions(width, height, spp, 1 if force_direct else 0, seed)
        if skies:
            self._sky_arrays = [np.ascontiguousarray(t, dtype=np.float32) for t in skies.values()]
            self._sky_structs = (abi.PrcSky * len(skies))(*[
                abi.PrcSky(None if name is None else name.encode(), t.ctypes.data_as(C.POINTER(C.c_float)), t.shape[1], t.shape[0])
                for name, t in zip(skies.keys(), self._sky_arrays)])
            opt.n_skies, opt.skies = len(skies), self._sky_structs
        h = C.c_void_p()
        if path is not None:
            rc = lib.prgpu_prc_load_file(os.fsencode(path), C.byref(opt), C.byref(h))
        else:
            rc = lib.prgpu_prc_load_string(source.encode(), os.fsencode(include_dir) if include_dir else None, C.byref(opt), C.byref(h))
        if rc != 0:
            raise abi.PrgpuError("prgpu error %d: %s" % (rc, lib.prgpu_prc_last_error().decode()), rc)
        self._lib, self._h = lib, h
        self.desc = lib.prgpu_prc_desc(h).contents
        self.warnings = [w for w in lib.prgpu_prc_warnings(h).decode().split("\n") if w]

    def sky_params(self):
        """{light index: (sun elevation, sun azimuth, turbidity, albedo[11])} of the scene's sky lights (what their SkyModel was built from)."""
        out = {}
        for i in range(self.desc.n_lights):
            if self.desc.lights[i].kind == abi.LIGHT_SKY:
                sp = abi.SkyParams()
                abi.check(self._lib.prgpu_prc_sky_info(self._h, i, C.byref(sp)))
                out[i] = (sp.sun_elevation, sp.sun_azimuth, sp.turbidity, [float(a) for a in sp.albedo])
        return out

    def outputs(self):
        """(channel array, count) of the scene's (output ...) blocks (OutputSpecification.cpp:254-365)."""
        n = C.c_uint32()
        ch = self._lib.prgpu_prc_outputs(self._h, C.byref(n))
        return ch, n.value

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.prgpu_prc_free(self._h)
            self._h = None

    settings = SceneData.settings
    width = SceneData.width
    height = SceneData.height
    spp = SceneData.spp


# ---- scene cache: a prgpu_scene_desc as one .npz file -----------------------------------------------------------

_STRUCT_ARRAYS = (("entities", "n_entities", abi.Entity), ("materials", "n_materials", abi.Material), ("emissions", "n_emissions", abi.Emission),
                  ("spectra", "n_spectra", abi.Spectrum), ("lights", "n_lights", abi.Light))


def hosek_sky_table(sun_elevation, sun_azimuth, turbidity=3.0, albedo=0.15, elevation_count=256, azimuth_count=512):
    """SkyModel::mData (src/skysun/skysun/SkyModel.cpp:15-56) through prgpu_sky_table: float32 [elevation, azimuth, 11]."""
    alb = (C.c_float * abi.SKY_BANDS)(*([albedo] * abi.SKY_BANDS if np.isscalar(albedo) else list(albedo)))
    t = np.zeros((elevation_count, azimuth_count, abi.SKY_BANDS), dtype=np.float32)
    rc = abi.load().prgpu_sky_table(sun_elevation, sun_azimuth, turbidity, alb, azimuth_count, elevation_count, t.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise abi.PrgpuError("prgpu_sky_table: error %d (turbidity must lie in 1 ... 10)" % rc)
    return t


def save_scene_npz(path, desc, drop_sky_tables=True, sky_params=None):
    """Write a scene description (from SceneBuilder or the .prc loader) as a compressed .npz.  The tables of SKY lights -- 5.8 MB each at
    the default resolution -- are left out by default; `sky_params` ({light index: (elevation, azimuth, turbidity, albedo[11])}, e.g.
    PrcScene.sky_params()) is stored instead, and ArrayScene rebuilds the tables from it (prgpu_sky_table)."""
    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n and ptr else np.zeros(0, dt)
    out = {"positions": arr(desc.positions, 3 * desc.n_vertices, np.float32), "normals": arr(desc.normals, 3 * desc.n_vertices, np.float32),
           "uvs": arr(desc.uvs, 2 * desc.n_vertices, np.float32), "indices": arr(desc.indices, 3 * desc.n_triangles, np.uint32),
           "tri_material": arr(desc.tri_material, desc.n_triangles, np.uint32),
           "camera": np.frombuffer(C.string_at(C.addressof(desc.camera), C.sizeof(abi.Camera)), dtype=np.uint8).copy(),
           "settings": np.frombuffer(C.string_at(C.addressof(desc.settings), C.sizeof(abi.Settings)), dtype=np.uint8).copy(),
           "api_version": np.array([desc.api_version], dtype=np.uint32)}
    tables = arr(desc.spectral_tables, desc.n_spectral_table_values, np.float32)
    lights = [abi.Light.from_buffer_copy(C.string_at(C.addressof(desc.lights[i]), C.sizeof(abi.Light))) for i in range(desc.n_lights)]
    spectra = [abi.Spectrum.from_buffer_copy(C.string_at(C.addressof(desc.spectra[i]), C.sizeof(abi.Spectrum))) for i in range(desc.n_spectra)]
    if drop_sky_tables:
        cuts = sorted((l.table_offset, l.azimuth_count * l.elevation_count * abi.SKY_BANDS) for l in lights if l.kind == abi.LIGHT_SKY)
        keep = np.ones(len(tables), dtype=bool)
        for off, n in cuts:
            keep[off:off + n] = False
        shift = lambda off: off - sum(n for o, n in cuts if o < off)  # noqa: E731
        for sp in spectra:
            if sp.kind in (abi.SPEC_TABLE, abi.SPEC_SELLMEIER):
                sp.table_offset = shift(sp.table_offset)
        tables = tables[keep]
    entities = [abi.Entity.from_buffer_copy(C.string_at(C.addressof(desc.entities[i]), C.sizeof(abi.Entity))) for i in range(desc.n_entities)]
    if drop_sky_tables:
        for en in entities:
            if en.kind == abi.ENTITY_QUADRIC:
                en.params = shift(en.params)
    out["tables"] = tables
    if sky_params:
        out["sky_params"] = np.array([[i, el, az, tu] + list(al) for i, (el, az, tu, al) in sorted(sky_params.items())], dtype=np.float32)
    for name, count, cls in _STRUCT_ARRAYS:
        n = getattr(desc, count)
        src = lights if name == "lights" else (spectra if name == "spectra" else (entities if name == "entities" else [getattr(desc, name)[i] for i in range(n)]))
        out[name] = np.frombuffer(b"".join(C.string_at(C.addressof(x), C.sizeof(cls)) for x in src), dtype=np.uint8).copy()
    np.savez_compressed(path, **out)


class ArrayScene:
    """A scene description rebuilt from save_scene_npz; quacks like SceneData (.desc, .settings, .width, .height, .spp)."""

    def __init__(self, path, sky_tables=None):
        z = np.load(path)
        # v6 caches load unchanged: prgpu_light kept its size, and the v6 prgpu_camera is a prefix of the v7 one (the new fields are zero)
        assert int(z["api_version"][0]) in (6, 7, abi.PRGPU_API_VERSION), "scene cache written for another ABI version"  # the structs kept their layout
        self.positions, self.indices, self.tri_material = z["positions"], z["indices"], z["tri_material"]
        self.normals = z["normals"] if len(z["normals"]) else None
        self.uvs = z["uvs"] if len(z["uvs"]) else None
        tables = [z["tables"]]
        self._structs = {}
        for name, count, cls in _STRUCT_ARRAYS:
            raw = z[name].tobytes()
            n = len(raw) // C.sizeof(cls)
            self._structs[name] = (cls * max(1, n)).from_buffer_copy(raw.ljust(C.sizeof(cls) * max(1, n), b"\0"))
            self._structs[name + "_n"] = n
        offset = len(tables[0])
        skies = list(sky_tables or [])
        params = {int(r[0]): r[1:] for r in z["sky_params"]} if "sky_params" in z.files else {}
        for i in range(self._structs["lights_n"]):
            l = self._structs["lights"][i]
            if l.kind == abi.LIGHT_SKY:
                if skies:
                    t = np.ascontiguousarray(skies.pop(0), dtype=np.float32)
                else:  # the Hosek-Wilkie table of the light's SkyModel, rebuilt from the stored parameters
                    if i not in params:
                        raise ValueError("%s: sky light %d has neither a table nor stored sky_params (a cache written before they were kept): "
                                         "pass sky_tables=[...] or regenerate the file with save_scene_npz(..., sky_params=...)" % (path, i))
                    el, az, tu = (float(v) for v in params[i][:3])
                    t = hosek_sky_table(el, az, tu, [float(a) for a in params[i][3:]], l.elevation_count, l.azimuth_count)
                assert t.shape == (l.elevation_count, l.azimuth_count, abi.SKY_BANDS), "sky table shape %s" % (t.shape,)
                l.table_offset = offset
                tables.append(t.reshape(-1))
                offset += t.size
        self.tables = np.ascontiguousarray(np.concatenate(tables), dtype=np.float32)
        if not len(self.tables):
            self.tables = np.zeros(1, np.float32)
        d = abi.SceneDesc()
        d.api_version = abi.PRGPU_API_VERSION
        f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        d.n_vertices = len(self.positions) // 3
        d.positions = self.positions.ctypes.data_as(f32p)
        d.normals = self.normals.ctypes.data_as(f32p) if self.normals is not None else None
        d.uvs = self.uvs.ctypes.data_as(f32p) if self.uvs is not None else None
        d.n_triangles = len(self.indices) // 3
        d.indices = self.indices.ctypes.data_as(u32p)
        d.tri_material = self.tri_material.ctypes.data_as(u32p)
        for name, count, cls in _STRUCT_ARRAYS:
            setattr(d, count, self._structs[name + "_n"])
            setattr(d, name, self._structs[name])
        d.n_spectral_table_values = offset
        d.spectral_tables = self.tables.ctypes.data_as(f32p)
        d.camera = abi.Camera.from_buffer_copy(z["camera"].tobytes().ljust(C.sizeof(abi.Camera), b"\0"))
        d.settings = abi.Settings.from_buffer_copy(z["settings"].tobytes())
        self.desc = d


for _p in ("settings", "width", "height", "spp"):
    setattr(ArrayScene, _p, getattr(SceneData, _p))


def load_prc(path, **overrides):
    """SceneLoader::loadFromFile for the supported part of the scene language."""
    return PrcScene(path=path, **overrides)


# ---- stock scenes (BASELINE.json configs) -----------------------------------------------------------------

def _cornell_into(b, data=None, material_override=None):
    if data is None:
        with open(os.path.join(_DATA, "cornell_box.json")) as f:
            data = json.load(f)
    cam = data["camera"]
    b.set_camera(np.asarray(cam["transform"], dtype=np.float32).reshape(4, 4), width=cam["width"][0], height=cam["height"][0],
                 near=cam["near"][0], far=cam["far"][0], local_direction=cam["local_direction"],
                 local_right=cam["local_right"], local_up=cam["local_up"])
    # block order of examples/cornellbox.prc (emission before the materials), so that ids equal those of the .prc loader
    radiance = b.smul(b.illuminant_d65(), b.illum(*data["emission"]["illum"]))  # (smul (illuminant "D65") (illum 17 12 4))
    ems = b.diffuse_emission(radiance)
    mats = {name: (material_override[name](b) if material_override and name in material_override else b.lambert(b.refl(*m["refl"])))
            for name, m in data["materials"].items()}
    for e in data["entities"]:
        b.add_mesh(e["p"], e["faces"], mats[e["material"]], normals=e["n"], emission=ems if e["emission"] else None,
                   transform=np.asarray(e["transform"], dtype=np.float32).reshape(4, 4))
    return mats


def cornell_box(width=256, height=256, spp=16, sampler=abi.SAMPLER_MJITT, **settings):
    """C1/C3: examples/cornellbox.prc geometry + materials with the `direct` integrator."""
    b = SceneBuilder(width, height)
    b.settings.aa_sampler, b.settings.aa_samples = sampler, spp
    for k, v in settings.items():
        setattr(b.settings, k, v)
    _cornell_into(b)
    return b.build()


def cornell_glassy(width=256, height=256, spp=16, ior="bk7", thin=False, tinted=False, sampler=abi.SAMPLER_MJITT, **settings):
    """The Cornell box with glass boxes (examples/cornellbox_glassy.prc's tallBox material on both boxes): smooth dielectric with a
    tabulated Sellmeier index (hero-wavelength collapse) or a constant index (`ior` a number)."""
    b = SceneBuilder(width, height)
    b.settings.aa_sampler, b.settings.aa_samples = sampler, spp
    for k, v in settings.items():
        setattr(b.settings, k, v)

    def glass(bb):
        n = bb.lookup_index(ior) if isinstance(ior, str) else bb.spectrum_const(float(ior))
        if tinted:
            return bb.dielectric(n, specularity=bb.refl(0.9, 0.9, 0.9), transmission=bb.refl(0.4, 0.8, 0.5), thin=thin)
        return bb.dielectric(n, thin=thin)

    _cornell_into(b, material_override={"shortBox": glass, "tallBox": glass})
    return b.build()


def cornell_metal(width=256, height=256, spp=16, sampler=abi.SAMPLER_MJITT, **settings):
    """The Cornell box with a default conductor on the tall box and a tinted, measured-like (tabulated eta/k) one on the short box."""
    b = SceneBuilder(width, height)
    b.settings.aa_sampler, b.settings.aa_samples = sampler, spp
    for k, v in settings.items():
        setattr(b.settings, k, v)
    gold = lambda bb: bb.conductor(eta=bb.spectrum_table(400.0, 700.0, [1.47, 1.38, 0.95, 0.35, 0.21, 0.16, 0.14]),
                                   k=bb.spectrum_table(400.0, 700.0, [1.95, 1.91, 1.85, 2.37, 2.88, 3.30, 3.70]), specularity=bb.refl(0.95, 0.95, 0.95))
    _cornell_into(b, material_override={"tallBox": lambda bb: bb.conductor(), "shortBox": gold})
    return b.build()


def pcg32_fast_floats(seed, n):
    """n floats in [0,1) from pcg32_fast(seed) (vectorised MCG: state_k = state_0 * M^k)."""
    mult = np.uint64(6364136223846793005)
    with np.errstate(over="ignore"):
        states = np.empty(n, dtype=np.uint64)
        states[0] = np.uint64(seed) | np.uint64(3)
        # cumulative products in blocks to stay vectorised
        block = 1 << 16
        pw = np.empty(block, dtype=np.uint64)
        pw[0] = np.uint64(1)
        pw[1:] = mult
        pw = np.multiply.accumulate(pw)
        jump = pw[-1] * mult
        base = states[0]
        for s in range(0, n, block):
            e = min(n, s + block)
            states[s:e] = base * pw[: e - s]
            base = base * jump
        x = states ^ (states >> np.uint64(22))
        out = (x >> (np.uint64(22) + (states >> np.uint64(61)))).astype(np.uint32)
    bits = (out >> np.uint32(9)) | np.uint32(0x3F800000)
    return bits.view(np.float32) - np.float32(1.0)


def triangle_soup(n, seed=42, size=0.01, lo=(-0.95, -0.95, 0.05), hi=(0.95, 0.95, 1.90)):
    """SURVEY 8(d) C4 soup: centroid ~ U(box), two edge vectors ~ U([-s,s]^3), vertices c, c+e1, c+e2."""
    u = pcg32_fast_floats(seed, 9 * n).reshape(n, 9)
    lo, hi = np.asarray(lo, dtype=np.float32), np.asarray(hi, dtype=np.float32)
    c = lo + u[:, 0:3] * (hi - lo)
    e1 = (u[:, 3:6] * np.float32(2) - np.float32(1)) * np.float32(size)
    e2 = (u[:, 6:9] * np.float32(2) - np.float32(1)) * np.float32(size)
    pos = np.stack([c, c + e1, c + e2], axis=1).reshape(-1, 3).astype(np.float32)
    faces = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return pos, faces


def cornell_rough(width=256, height=256, spp=16, roughness=0.2, vndf=True, sampler=abi.SAMPLER_MJITT, **settings):
    """The Cornell box with GGX materials: a rough gold-like conductor (short box), rough tinted BK7 glass (tall box) and an anisotropic
    rough conductor (left wall) -- roughconductor.cpp / roughdielectric.cpp closures under NEE + MIS."""
    b = SceneBuilder(width, height)
    b.settings.aa_sampler, b.settings.aa_samples = sampler, spp
    for k, v in settings.items():
        setattr(b.settings, k, v)
    _cornell_into(b, material_override={
        "shortBox": lambda bb: bb.rough_conductor(roughness, eta=bb.spectrum_const(0.2), k=bb.spectrum_const(3.9), vndf=vndf),
        "tallBox": lambda bb: bb.rough_dielectric(0.75 * roughness, ior=bb.lookup_index("bk7"), transmission=bb.refl(0.6, 0.9, 0.7), vndf=vndf),
        "leftWall": lambda bb: (bb.rough_conductor(1.5 * roughness, roughness_y=0.5 * roughness, specularity=bb.refl(0.9, 0.6, 0.3)) if vndf
                                else bb.rough_conductor(1.5 * roughness, specularity=bb.refl(0.9, 0.6, 0.3), vndf=False))})
    return b.build()


def cornell_soup(width=1920, height=1080, spp=1024, n_triangles=1_000_000, sampler=abi.SAMPLER_SOBOL, **settings):
    """C4 (headline): the 32 Cornell triangles + (n_triangles - 32) soup triangles with the white Lambert."""
    b = SceneBuilder(width, height)
    b.settings.aa_sampler, b.settings.aa_samples = sampler, spp
    for k, v in settings.items():
        setattr(b.settings, k, v)
    mats = _cornell_into(b)
    n_soup = max(0, n_triangles - 32)
    if n_soup:
        pos, faces = triangle_soup(n_soup)
        b.add_mesh(pos, faces, mats["backWall"])
    return b.build()


def sphere_light(width=512, height=512, spp=64, **settings):
    """C2: tessellated unit sphere (64x128 lat-long) on a 10x10 ground quad under a 1x1 D65x10 area light."""
    b = SceneBuilder(width, height)
    s = b.settings
    s.aa_sampler, s.aa_samples, s.filter, s.filter_radius, s.mapper = abi.SAMPLER_MJITT, spp, abi.FILTER_BLOCK, 0, abi.MAPPER_RANDOM
    for k, v in settings.items():
        setattr(s, k, v)
    grey = b.lambert(b.spectrum_const(0.8))
    ems = b.diffuse_emission(b.smul(b.illuminant_d65(), b.spectrum_const(10.0)))
    nlat, nlon = 64, 128
    th = np.linspace(0, np.pi, nlat + 1, dtype=np.float64)
    ph = np.linspace(0, 2 * np.pi, nlon, endpoint=False, dtype=np.float64)
    P = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)), np.outer(np.cos(th), np.ones_like(ph))], axis=-1)
    pos = P.reshape(-1, 3).astype(np.float32)
    faces = []
    for i in range(nlat):
        for j in range(nlon):
            a, bb = i * nlon + j, i * nlon + (j + 1) % nlon
            c, d = (i + 1) * nlon + j, (i + 1) * nlon + (j + 1) % nlon
            if i != 0:
                faces.append([a, c, bb])
            if i != nlat - 1:
                faces.append([bb, c, d])
    b.add_mesh(pos, faces, grey, normals=pos.copy())
    b.add_mesh([[-5, -5, -1], [5, -5, -1], [5, 5, -1], [-5, 5, -1]], [[0, 1, 2], [0, 2, 3]], grey)
    b.add_mesh([[-0.5, -0.5, 3], [-0.5, 0.5, 3], [0.5, 0.5, 3], [0.5, -0.5, 3]], [[0, 1, 2], [0, 2, 3]], grey, emission=ems)
    # camera at (0,-4,1.5) looking at the origin
    eye = np.array([0, -4, 1.5]); fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, [0, 0, 1]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    T = np.eye(4, dtype=np.float32)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = right, up, fwd, eye
    b.set_camera(T, width=0.72, height=0.72)
    return b.build()


def cbox_eval(width=256, height=256, spp=128, **settings):
    """examples/evaluation/scene.prc (Mitsuba-2 comparison scene): measured spectra, quads, sobol 128, depth 6."""
    with open(os.path.join(_DATA, "cbox_eval.json")) as f:
        data = json.load(f)
    b = SceneBuilder(width, height)
    s = b.settings
    s.aa_sampler, s.aa_samples, s.max_ray_depth, s.mapper, s.filter, s.filter_radius = abi.SAMPLER_SOBOL, spp, 6, abi.MAPPER_RANDOM, abi.FILTER_TRIANGLE, 0
    for k, v in settings.items():
        setattr(s, k, v)
    mats = {n: b.lambert(b.spectrum_table(m["start"], m["end"], m["values"])) for n, m in data["materials"].items()}
    e = data["emission"]
    ems = b.diffuse_emission(b.spectrum_table(e["start"], e["end"], e["values"]))
    for ent in data["entities"]:
        T = np.eye(4, dtype=np.float32)
        if ent["position"]:
            T[:3, 3] = ent["position"]
        b.add_mesh(ent["p"], ent["faces"], mats[ent["material"]], normals=ent.get("n"), emission=ems if ent["emission"] else None, transform=T)
    cam = data["camera"]
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = cam["position"]
    b.set_camera(T, width=cam["width"][0], height=cam["height"][0], near=cam["near"][0], far=cam["far"][0],
                 local_direction=cam["local_direction"], local_right=cam["local_right"], local_up=cam["local_up"])
    return b.build()
