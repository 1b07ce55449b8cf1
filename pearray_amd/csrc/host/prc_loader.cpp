// prc_loader.cpp -- ".prc" scene files -> prgpu_scene_desc (include/prgpu.h: prgpu_prc_*).
//
// Replaces, for the part of PearRay the `direct` hot path evaluates, what SceneLoader does between the parsed DataLisp tree and
// the Scene/RenderContext (reference file:line cited at each block):
//   top level `(scene ...)`                                        SceneLoader.cpp:73-160
//   dispatch of inner blocks                                       SceneLoader.cpp:162-198
//   sampler / filter / spectral_mapper / integrator blocks         SceneLoader.cpp:192-384 + the plugins' getNames()/parameters
//   :transform / :position :rotation :scale                        SceneLoader.cpp:386-444, parser/MathParser.cpp
//   camera (perspective)                                           plugins/main/cameras/perspective.cpp:141-165
//   material (diffuse/lambert), emission (diffuse)                 materials/lambert.cpp:90-117, emissions/diffuse.cpp:55-80
//   spectral expressions                                           SceneLoader.cpp:999-1040, node/SpectralValueNode.cpp:12-59,
//                                                                  IlluminantNode.cpp:86-130, SpectralConstNode.cpp:12-40, SpectralMathNode.cpp:265-280
//   inline meshes                                                  parser/MeshParser.cpp:11-255
//   entities of type 'mesh'                                        plugins/main/entities/mesh.cpp:205-330
//   (include "file")                                               SceneLoader.cpp:848-886
// Anything else the reference would accept but this backend cannot render (other entity, material, light, camera, sampler
// types ...) is an error (PRGPU_EUNSUPPORTED) naming the block -- never silently dropped; blocks that do not influence the
// radiance (outputs) are skipped with a warning.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/prgpu.h"
#include "../tables/pr_tables.inl"
#include "../tables/pr_illuminants.inl"
#include <zlib.h>

#include "datalisp.h"
#include "setup.h"

using prgpu_host::dl::Group;
using prgpu_host::dl::Value;

struct prgpu_prc {
	std::vector<float> positions, normals, uvs, tables;
	std::vector<uint32_t> indices, tri_material;
	std::vector<prgpu_entity> entities;
	std::vector<prgpu_material> materials;
	std::vector<prgpu_emission> emissions;
	std::vector<prgpu_spectrum> spectra;
	std::vector<prgpu_light> lights;
	std::vector<prgpu_sky_params> sky_params; // per light (sky lights only: what SkyModel was built from)
	std::vector<prgpu_output_channel> outputs;
	std::vector<std::string> output_names;
	prgpu_scene_desc desc;
	std::string warnings;
};

namespace {

thread_local std::string g_prc_error;

struct LoadError {
	int code;
	std::string msg;
};
[[noreturn]] void fail(int code, const std::string& msg) { throw LoadError{ code, msg }; }
std::string where(const Group& g) { return "(" + g.id + " ...) at line " + std::to_string(g.line); }
std::string lower(std::string s)
{
	std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
	return s;
}

struct Mesh {
	std::vector<float> p, n, uv; // uv: 2 per vertex (`t` / `uv` attribute, MeshParser.cpp:163-167)
	std::vector<std::vector<uint32_t>> faces;
	std::vector<uint32_t> slots; // per face material slot (empty: slot 0)
};

struct Loader {
	prgpu_prc& out;
	prgpu_prc_options opt;
	prgpu_settings settings;
	std::map<std::string, uint32_t> material_ids, emission_ids;
	std::map<std::string, Mesh> meshes;
	std::map<std::string, prgpu_camera> cameras;
	std::string first_camera, selected_camera;
	bool any_normals = false, any_uvs = false, have_integrator = false, have_filter = false;
	int include_depth = 0;

	Loader(prgpu_prc& o, const prgpu_prc_options* op) : out(o)
	{
		std::memset(&opt, 0, sizeof(opt));
		if (op)
			opt = *op;
		prgpu_settings_default(&settings);
	}
	void warn(const std::string& m) { out.warnings += m + "\n"; }

	// ---- small accessors -----------------------------------------------------------------------------------
	static std::string get_string(const Group& g, const char* key, const std::string& def)
	{
		const Value* v = g.get(key);
		return v && v->type == Value::STRING ? v->s : def;
	}
	static double get_number(const Group& g, const char* key, double def)
	{
		const Value* v = g.get(key);
		if (v && v->type == Value::GROUP && (v->g->id == "deg2rad" || v->g->id == "rad2deg") && v->g->anonymous_count() == 1 && v->g->at(0).is_number()) {
			// SceneLoader::unpackShadingNetwork (SceneLoader.cpp:1025-1034): (deg2rad x) = x * PR_DEG2RAD in fp32
			const float pi = 3.14159265358979323846f;
			return v->g->id == "deg2rad" ? (float)v->g->at(0).number() * (pi / 180.0f) : (float)v->g->at(0).number() * (180.0f / pi);
		}
		return v && v->is_number() ? v->number() : def;
	}
	static bool get_bool(const Group& g, const char* key, bool def)
	{
		const Value* v = g.get(key);
		return v && v->type == Value::BOOL ? v->b : def;
	}
	static bool get_vec3(const Group& g, const char* key, float dst[3]) // MathParser::getVector: 2 or 3 numbers
	{
		const Value* v = g.get(key);
		if (!v || v->type != Value::GROUP || !v->g->all_numbers())
			return false;
		const size_t n = v->g->anonymous_count();
		if (n != 2 && n != 3)
			return false;
		for (size_t i = 0; i < 3; ++i)
			dst[i] = i < n ? (float)v->g->at(i).number() : 0.0f;
		return true;
	}

	// ---- transforms (SceneLoader.cpp:386-444) -----------------------------------------------------------------
	static void identity(float m[16])
	{
		for (int i = 0; i < 16; ++i)
			m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
	}
	void transform_of(const Group& g, float m[16])
	{
		identity(m);
		const Value* t = g.get("transform");
		if (t && t->type == Value::GROUP) {
			const Group& a = *t->g;
			if (a.all_numbers() && a.anonymous_count() == 16) {
				for (int i = 0; i < 16; ++i)
					m[i] = (float)a.at(i).number();
			} else if (a.all_numbers() && a.anonymous_count() == 9) {
				for (int i = 0; i < 3; ++i)
					for (int j = 0; j < 3; ++j)
						m[4 * i + j] = (float)a.at(3 * i + j).number();
			} else {
				fail(PRGPU_EINVAL, ":transform of " + where(g) + " must have 16 or 9 numbers");
			}
			// Transformf::makeAffine: last row becomes 0 0 0 1
			m[12] = m[13] = m[14] = 0.0f;
			m[15]				  = 1.0f;
			return;
		}
		float pos[3] = { 0, 0, 0 }, sca[3] = { 1, 1, 1 };
		float q[4] = { 1, 0, 0, 0 }; // w x y z
		if (g.get("position") && !get_vec3(g, "position", pos))
			fail(PRGPU_EINVAL, ":position of " + where(g) + " is not a vector");
		if (const Value* r = g.get("rotation")) {
			bool ok = false;
			if (r->type == Value::GROUP) {
				const Group& a = *r->g;
				if (a.is_array && a.anonymous_count() == 4 && a.all_numbers()) {
					for (int i = 0; i < 4; ++i)
						q[i] = (float)a.at(i).number();
					ok = true;
				} else if (a.id == "euler" && a.anonymous_count() == 3 && a.all_numbers()) { // degrees, applied z*y*x
					const double x = a.at(0).number() * M_PI / 180 / 2, y = a.at(1).number() * M_PI / 180 / 2, z = a.at(2).number() * M_PI / 180 / 2;
					const double qx[4] = { std::cos(x), std::sin(x), 0, 0 }, qy[4] = { std::cos(y), 0, std::sin(y), 0 }, qz[4] = { std::cos(z), 0, 0, std::sin(z) };
					auto mul = [](const double a[4], const double b[4], double r[4]) {
						r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
						r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
						r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
						r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
					};
					double zy[4], zyx[4];
					mul(qz, qy, zy);
					mul(zy, qx, zyx);
					for (int i = 0; i < 4; ++i)
						q[i] = (float)zyx[i];
					ok = true;
				}
			}
			if (!ok)
				fail(PRGPU_EINVAL, ":rotation of " + where(g) + " must be a quaternion [w,x,y,z] or (euler x y z)");
		}
		if (const Value* s = g.get("scale")) {
			if (s->is_number())
				sca[0] = sca[1] = sca[2] = (float)s->number();
			else if (!get_vec3(g, "scale", sca))
				fail(PRGPU_EINVAL, ":scale of " + where(g) + " is not a number or vector");
		}
		const float qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
		if (!(qn > 0))
			fail(PRGPU_EINVAL, ":rotation of " + where(g) + " has zero length");
		const float w = q[0] / qn, x = q[1] / qn, y = q[2] / qn, z = q[3] / qn;
		const float R[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
							 2 * (y * z - x * w),	  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y) };
		for (int i = 0; i < 3; ++i) {
			for (int j = 0; j < 3; ++j)
				m[4 * i + j] = R[3 * i + j] * sca[j]; // T * R * S
			m[4 * i + 3] = pos[i];
		}
	}

	// ---- spectral expressions ----------------------------------------------------------------------------------
	uint32_t add_spectrum(const prgpu_spectrum& s)
	{
		out.spectra.push_back(s);
		return (uint32_t)out.spectra.size() - 1;
	}
	static prgpu_spectrum blank(uint32_t kind)
	{
		prgpu_spectrum s;
		std::memset(&s, 0, sizeof(s));
		s.kind = kind;
		return s;
	}
	uint32_t spectrum_const(float v)
	{
		prgpu_spectrum s = blank(PRGPU_SPEC_CONST);
		s.p[0]			 = v;
		return add_spectrum(s);
	}
	uint32_t spectrum_table(float start, float end, const float* values, size_t n)
	{
		prgpu_spectrum s = blank(PRGPU_SPEC_TABLE);
		s.table_offset	 = (uint32_t)out.tables.size();
		s.table_count	 = (uint32_t)n;
		s.wl_start		 = start;
		s.wl_end		 = end;
		for (size_t i = 0; i < n; ++i)
			out.tables.push_back(std::max(0.0f, values[i]));
		return add_spectrum(s);
	}
	uint32_t spectrum_sellmeier(const float* b, const float* c, int n)
	{
		prgpu_spectrum s = blank(PRGPU_SPEC_SELLMEIER);
		s.table_offset	 = (uint32_t)out.tables.size();
		s.table_count	 = (uint32_t)(2 * n);
		out.tables.insert(out.tables.end(), b, b + n);
		out.tables.insert(out.tables.end(), c, c + n);
		return add_spectrum(s);
	}
	// a value in a spectral slot: number | (refl r g b) | (illum r g b) | (illuminant "D65") | (spectrum ...) | (smul a b)
	uint32_t spectral_node(const Value& v, const Group& owner, const char* key)
	{
		if (v.is_number())
			return spectrum_const((float)v.number());
		if (v.type == Value::STRING) {
			// SceneLoadContext::lookupSpectralNode (SceneLoadContext.cpp:222-232): a string names a (node ...)/(texture ...) block; an
			// unknown name silently yields the parameter's default.  Named blocks are rejected when they are declared (dispatch), so
			// every name is unknown here.
			warn(std::string(":") + key + " of " + where(owner) + " names the unknown node '" + v.s + "': default value used (as the reference does)");
			return spectrum_const(string_default);
		}
		if (v.type != Value::GROUP || v.g->is_array)
			fail(PRGPU_EINVAL, std::string(":") + key + " of " + where(owner) + " is not a spectral expression");
		const Group& e		 = *v.g;
		const std::string id = lower(e.id);
		if (id == "refl" || id == "reflection" || id == "illum" || id == "illumination") {
			float rgb[3];
			for (int k = 0; k < 3; ++k) {
				rgb[k] = e.at(k).is_number() ? (float)e.at(k).number() : 0.0f;
				if (!(rgb[k] >= 0.0f) || !std::isfinite(rgb[k]))
					fail(PRGPU_EINVAL, where(e) + ": colour components must be finite and non-negative");
			}
			prgpu_spectrum s;
			if (id[0] == 'r') { // SpectralValueNode.cpp:16-30
				s = blank(PRGPU_SPEC_PARAMETRIC);
				prgpu_host::rgb_to_coeffs(rgb, s.p);
			} else { // :31-47: scaled so that the fitted colour has maximum 0.5
				s				= blank(PRGPU_SPEC_PARAMETRIC_SCALED);
				const float mx	= std::max(rgb[0], std::max(rgb[1], rgb[2]));
				float power		= 1.0f;
				float scaled[3] = { rgb[0], rgb[1], rgb[2] };
				if (mx > 0.0f) {
					const float scale = 2 * mx;
					for (int k = 0; k < 3; ++k)
						scaled[k] = rgb[k] / scale;
					power = scale;
				}
				prgpu_host::rgb_to_coeffs(scaled, s.p);
				s.p[3] = power;
			}
			return add_spectrum(s);
		}
		if (id == "illuminant") { // IlluminantNode.cpp:89-130
			std::string name = get_string(e, "spectrum", "");
			if (name.empty())
				name = e.at(0).type == Value::STRING ? e.at(0).s : "D65";
			name = lower(name);
			if (name == "d65")
				return spectrum_table(300.0f, 830.0f, PR_D65, 107);
			if (name == "e")
				return spectrum_const(1.0f);
			static const struct {
				const char* name;
				const float* data;
			} wide[] = { { "a", PR_ILL_A }, { "c", PR_ILL_C }, { "d50", PR_ILL_D50 }, { "d55", PR_ILL_D55 }, { "d75", PR_ILL_D75 } },
			  fluorescent[] = { { "f1", PR_ILL_F1 }, { "f2", PR_ILL_F2 }, { "f3", PR_ILL_F3 }, { "f4", PR_ILL_F4 }, { "f5", PR_ILL_F5 }, { "f6", PR_ILL_F6 },
								{ "f7", PR_ILL_F7 }, { "f8", PR_ILL_F8 }, { "f9", PR_ILL_F9 }, { "f10", PR_ILL_F10 }, { "f11", PR_ILL_F11 }, { "f12", PR_ILL_F12 } };
			for (const auto& w : wide) // CIE_SampleCount samples over [300, 830] nm (IlluminantData.inl:4-6)
				if (name == w.name)
					return spectrum_table(300.0f, 830.0f, w.data, 107);
			for (const auto& f : fluorescent) // CIE_F_SampleCount samples over [380, 780] nm (:8-10)
				if (name == f.name)
					return spectrum_table(380.0f, 780.0f, f.data, 81);
			fail(PRGPU_EINVAL, where(e) + ": unknown illuminant '" + name + "'");
		}
		if (id == "spectrum") { // SpectralConstNode.cpp:12-33
			const float start = (float)get_number(e, "start", 0.0), end = (float)get_number(e, "end", 0.0);
			if (!(start < end))
				fail(PRGPU_EINVAL, where(e) + ": invalid :start/:end wavelengths");
			std::vector<float> vals;
			for (size_t i = 0; i < e.anonymous_count(); ++i)
				vals.push_back(e.at(i).is_number() ? (float)e.at(i).number() : 0.0f);
			if (vals.size() < 2)
				fail(PRGPU_EINVAL, where(e) + ": a spectrum needs at least two values");
			return spectrum_table(start, end, vals.data(), vals.size());
		}
		if (id == "checkerboard" || id == "grid") { // CheckerboardNode.cpp:78-90: (checkerboard a b [su [sv]])
			const size_t n	 = e.anonymous_count();
			prgpu_spectrum s = blank(PRGPU_SPEC_CHECKER);
			string_default	 = 0.8f;
			s.lhs			 = n > 0 ? spectral_node(e.at(0), e, "op1") : spectrum_const(0.8f);
			string_default	 = 0.2f;
			s.rhs			 = n > 1 ? spectral_node(e.at(1), e, "op2") : spectrum_const(0.2f);
			for (size_t k = 2; k < 4; ++k)
				if (n > k && !e.at(k).is_number())
					fail(PRGPU_EUNSUPPORTED, where(e) + ": checkerboard scales must be numbers");
			s.p[0] = n > 2 ? (float)e.at(2).number() : 5.0f;
			s.p[1] = n > 3 ? (float)e.at(3).number() : s.p[0];
			s.p[2] = n <= 2 ? 0.0f : (n == 3 ? 1.0f : 2.0f);
			return add_spectrum(s);
		}
		if (id == "smul") { // SpectralMathNode.cpp:275: product of two spectral nodes
			if (e.anonymous_count() != 2)
				fail(PRGPU_EINVAL, where(e) + ": smul takes two operands");
			prgpu_spectrum s = blank(PRGPU_SPEC_MUL);
			s.lhs			 = spectral_node(e.at(0), e, "lhs");
			s.rhs			 = spectral_node(e.at(1), e, "rhs");
			if (out.spectra[s.lhs].kind == PRGPU_SPEC_MUL || out.spectra[s.rhs].kind == PRGPU_SPEC_MUL)
				fail(PRGPU_EUNSUPPORTED, where(e) + ": nested smul is not supported (operands must be leaves)");
			if (out.spectra[s.lhs].kind == PRGPU_SPEC_CHECKER || out.spectra[s.rhs].kind == PRGPU_SPEC_CHECKER)
				fail(PRGPU_EUNSUPPORTED, where(e) + ": a checkerboard inside smul is not supported");
			return add_spectrum(s);
		}
		if (id == "lookup_index") { // ReflectiveNode.cpp:224-246,387-396: tabulated Sellmeier coefficients (refractiveindex.info)
			std::string name = lower(e.at(0).type == Value::STRING ? e.at(0).s : get_string(e, "name", "bk7"));
			static const struct {
				const char* name;
				int n;
				float b[4], c[4];
			} table[] = {
				{ "bk7", 3, { 1.03961212f, 0.231792344f, 1.01046945f }, { 0.00600069867f, 0.0200179144f, 103.560653f } },
				{ "glass", 3, { 1.03961212f, 0.231792344f, 1.01046945f }, { 0.00600069867f, 0.0200179144f, 103.560653f } },
				{ "h2o", 4, { 5.684027565e-1f, 1.726177391e-1f, 2.086189578e-2f, 1.130748688e-1f }, { 5.101829712e-3f, 1.821153936e-2f, 2.620722293e-2f, 1.069792721e1f } },
				{ "water", 4, { 5.684027565e-1f, 1.726177391e-1f, 2.086189578e-2f, 1.130748688e-1f }, { 5.101829712e-3f, 1.821153936e-2f, 2.620722293e-2f, 1.069792721e1f } },
				{ "diamond", 2, { 0.3306f, 4.3356f }, { 0.030625f, 0.011236f } },
			};
			if (name == "vacuum" || name == "none")
				return spectrum_const(1.0f);
			if (name == "air")
				return spectrum_const(1.000277f);
			for (const auto& t : table)
				if (name == t.name)
					return spectrum_sellmeier(t.b, t.c, t.n);
			fail(PRGPU_EINVAL, where(e) + ": unknown lookup_index '" + name + "'");
		}
		if (id == "sellmeier_index") { // (sellmeier_index b1 c1 b2 c2 ...), ReflectiveNode.cpp:330-360 (constant coefficients only)
			const size_t n = e.anonymous_count();
			if (n < 2 || n > 8 || (n & 1) || !e.all_numbers())
				fail(PRGPU_EUNSUPPORTED, where(e) + ": sellmeier_index takes 1..4 constant (B, C) pairs");
			float b[4], c[4];
			for (size_t i = 0; i < n / 2; ++i) {
				b[i] = (float)e.at(2 * i).number();
				c[i] = (float)e.at(2 * i + 1).number();
			}
			return spectrum_sellmeier(b, c, (int)(n / 2));
		}
		if (id == "spd") // SPDFilePlugin::create (node/SPDFileNode.cpp:20-91): an equidistant spectrum from a CSV file
			return spectrum_spd_file(e);
		fail(PRGPU_EUNSUPPORTED, "spectral expression " + where(e) + " is not supported (number, refl, illum, illuminant, spectrum, spd, smul, lookup_index, checkerboard are)");
	}
	// CSV::read (base/container/CSV.cpp:52-163): ',' or ';' separate, empty tokens are skipped, a first row with a non-number is the
	// header, rows with another token count are dropped, tokens that are not numbers count as 0
	static bool read_csv(const std::string& path, std::vector<float>& data, size_t& columns)
	{
		std::ifstream f(path);
		if (!f)
			return false;
		auto split = [](const std::string& line, std::vector<std::string>& tokens, size_t limit) {
			tokens.clear();
			std::string token;
			for (const char c : line) {
				if (c == ',' || c == ';') {
					if (!token.empty())
						tokens.push_back(token);
					token.clear();
				} else {
					token += c;
				}
				if (limit && tokens.size() == limit)
					return;
			}
			if (!token.empty())
				tokens.push_back(token);
		};
		auto number = [](const std::string& t, float& v) { // std::stof in the "C" locale: leading blanks, the longest valid prefix
			char* end = nullptr;
			errno	  = 0;
			const float r = std::strtof(t.c_str(), &end);
			if (end == t.c_str())
				return false;
			v = (errno == ERANGE && !std::isfinite(r)) ? 0.0f : r; // out of range: silently 0 (CSV.cpp:143-146)
			return true;
		};
		std::string line;
		std::getline(f, line);
		if (!line.empty() && line.back() == '\r')
			line.pop_back();
		std::vector<std::string> tokens;
		split(line, tokens, 0);
		columns = tokens.size();
		if (columns == 0)
			return false;
		data.clear();
		bool header = false;
		for (const std::string& t : tokens) {
			float v;
			if (!number(t, v))
				header = true;
		}
		if (!header)
			for (const std::string& t : tokens) {
				float v = 0;
				(void)number(t, v);
				data.push_back(v);
			}
		while (std::getline(f, line)) {
			if (!line.empty() && line.back() == '\r')
				line.pop_back();
			split(line, tokens, columns);
			if (tokens.size() != columns)
				continue; // malformed line: skipped
			for (const std::string& t : tokens) {
				float v = 0;
				if (!number(t, v))
					v = 0.0f;
				data.push_back(v);
			}
		}
		return true;
	}
	uint32_t spectrum_spd_file(const Group& e)
	{
		std::string file = get_string(e, "file", "");
		if (file.empty() && e.anonymous_count() > 0 && e.at(0).type == Value::STRING)
			file = e.at(0).s;
		if (file.empty())
			fail(PRGPU_EINVAL, where(e) + ": no file given for spd");
		std::string path = file;
		if (path[0] != '/' && !current_dir.empty())
			path = current_dir + "/" + path;
		std::vector<float> data;
		size_t columns = 0;
		if (!read_csv(path, data, columns))
			fail(PRGPU_EIO, where(e) + ": spd file '" + path + "' is not valid");
		const size_t rows = data.size() / columns;
		if (columns <= 1 || rows < 2)
			fail(PRGPU_EINVAL, where(e) + ": spd file '" + file + "' has not enough data");
		const float delta = data[columns] - data[0], start = data[0], end = data[(rows - 1) * columns];
		if (delta <= 1.1920929e-07f)
			fail(PRGPU_EINVAL, where(e) + ": spd file '" + file + "' has invalid wavelengths");
		size_t column = 1;
		if (const Value* v = e.get("column"))
			if (v->is_number())
				column = (size_t)v->number();
		if (e.anonymous_count() > 1 && e.at(1).is_number())
			column = (size_t)e.at(1).number();
		if (column >= columns) // the reference clamps to columnCount and then reads past the row (SPDFileNode.cpp:66,82): refused here
			fail(PRGPU_EINVAL, where(e) + ": spd column " + std::to_string(column) + " does not exist in '" + file + "'");
		float norm = 1.0f;
		if (const Value* v = e.get("percentage"))
			if (v->type == Value::BOOL)
				norm = v->b ? (1 / 100.0f) : 1.0f;
		if (e.anonymous_count() > 2 && e.at(2).type == Value::BOOL)
			norm = e.at(2).b ? (1 / 100.0f) : 1.0f;
		std::vector<float> values(rows);
		bool uneven = false;
		for (size_t i = 0; i < rows; ++i) {
			if (std::fabs(data[i * columns] - (i * delta + start)) > 1.1920929e-07f)
				uneven = true;
			values[i] = data[i * columns + column] * norm; // spectrum_table clamps at zero like SPDFileNode.cpp:82
		}
		if (uneven)
			warn(where(e) + ": spd file '" + file + "' has non equidistant data (read as equidistant, as in the reference)");
		return spectrum_table(start, end, values.data(), rows);
	}
	std::string current_dir; // directory of the file being loaded (relative file names of spd nodes)
	float string_default = 1.0f; // default of the parameter being parsed (for strings naming unknown nodes)
	// FloatSpectralNode::eval of a loaded node at one wavelength (what SkyModel asks of the ground albedo, SkyModel.cpp:30-34): the
	// arithmetic of spectrum_leaf / spectrum_eval on the device
	float eval_spectrum_at(uint32_t id, float wl) const
	{
		auto leaf = [&](const prgpu_spectrum& n) -> float {
			switch (n.kind) {
			case PRGPU_SPEC_CONST: return n.p[0];
			case PRGPU_SPEC_PARAMETRIC:
			case PRGPU_SPEC_PARAMETRIC_SCALED: {
				const float x = (n.p[0] * wl + n.p[1]) * wl + n.p[2];
				const float v = (0.5f * x) * (1.0f / std::sqrt(x * x + 1.0f)) + 0.5f;
				return n.kind == PRGPU_SPEC_PARAMETRIC ? v : v * n.p[3];
			}
			case PRGPU_SPEC_TABLE: {
				const float delta = (n.wl_end - n.wl_start) / (n.table_count - 1);
				const float af	  = std::max(0.0f, (wl - n.wl_start) / delta);
				const int index	  = (int)std::min<float>(float(n.table_count - 2), af);
				const float t	  = std::min<float>(float(n.table_count - 1), af) - index;
				const float* data = out.tables.data() + n.table_offset;
				return data[index] * (1 - t) + data[index + 1] * t;
			}
			case PRGPU_SPEC_SELLMEIER: { // Scattering::sellmeier (base/math/Scattering.h:219-242), as spectrum_leaf on the device
				const uint32_t nc = n.table_count / 2;
				const float* B	  = out.tables.data() + n.table_offset;
				const float* C	  = B + nc;
				const float lq	  = wl / 1000;
				const float lq2	  = lq * lq;
				float value		  = 1.0f;
				for (uint32_t i = 0; i < nc; ++i)
					value += B[i] * lq2 / (lq2 - C[i]);
				return std::sqrt(value);
			}
			default: return 0.0f;
			}
		};
		// a checkerboard is asked at UV (0, 0), where the sky model's ShadingContext stands (SkyModel.cpp:30-34): the even cell, i.e. the
		// operand resolve_texture (render.hip) picks there
		for (int guard = 0; id < out.spectra.size() && out.spectra[id].kind == PRGPU_SPEC_CHECKER && guard < 64; ++guard)
			id = out.spectra[id].rhs;
		if (id >= out.spectra.size())
			return 0.0f;
		const prgpu_spectrum& n = out.spectra[id];
		if (n.kind == PRGPU_SPEC_MUL && n.lhs < out.spectra.size() && n.rhs < out.spectra.size())
			return leaf(out.spectra[n.lhs]) * leaf(out.spectra[n.rhs]);
		return leaf(n);
	}
	uint32_t spectral_param(const Group& g, std::initializer_list<const char*> keys, float def)
	{
		string_default = def;
		for (const char* k : keys)
			if (const Value* v = g.get(k))
				return spectral_node(*v, g, k);
		return spectrum_const(def);
	}

	// ---- blocks -------------------------------------------------------------------------------------------------
	void add_sampler(const Group& g) // SceneLoader.cpp:192-258
	{
		const std::string type = lower(get_string(g, "type", ""));
		const std::string slot = lower(get_string(g, "slot", "aa"));
		if (type.empty())
			fail(PRGPU_EINVAL, where(g) + ": no valid type given");
		uint32_t kind;
		static const char* mj[] = { "multijittered", "multi_jittered", "jittered", "multijitter", "multi_jitter", "jitter", "mjitt", "jitt" };
		if (type == "random")
			kind = PRGPU_SAMPLER_RANDOM;
		else if (type == "sobol")
			kind = PRGPU_SAMPLER_SOBOL;
		else if (std::find_if(std::begin(mj), std::end(mj), [&](const char* n) { return type == n; }) != std::end(mj))
			kind = PRGPU_SAMPLER_MJITT;
		else if (type == "uniform")
			kind = PRGPU_SAMPLER_UNIFORM;
		else if (type == "stratified")
			kind = PRGPU_SAMPLER_STRATIFIED;
		else if (type == "halton")
			kind = PRGPU_SAMPLER_HALTON;
		else if (type == "hammersley")
			kind = PRGPU_SAMPLER_HAMMERSLEY;
		else
			fail(PRGPU_EUNSUPPORTED, where(g) + ": sampler type '" + type + "' is not supported (random, uniform, stratified, mjitt, sobol, halton, hammersley are)");
		const uint32_t count = (uint32_t)std::max(1.0, get_number(g, "sample_count", 128)); // DEF_SAMPLE_COUNT
		if (slot == "aa" || slot == "pixel" || slot == "antialiasing") {
			settings.aa_sampler = kind;
			settings.aa_samples = count;
			settings.aa_base_x = settings.aa_base_y = settings.aa_burnin = 0;
			if (kind == PRGPU_SAMPLER_HALTON || kind == PRGPU_SAMPLER_HAMMERSLEY) { // HaltonSampler.cpp:169-192
				settings.aa_base_x = (uint32_t)get_number(g, "base_x", 13);
				settings.aa_base_y = (uint32_t)get_number(g, "base_y", 47);
				settings.aa_burnin = (uint32_t)get_number(g, "burnin", kind == PRGPU_SAMPLER_HALTON ? std::max(settings.aa_base_x, settings.aa_base_y) : settings.aa_base_x);
				if (settings.aa_base_x < 2 || settings.aa_base_y < 2)
					fail(PRGPU_EINVAL, where(g) + ": halton bases must be >= 2");
			} else if (kind == PRGPU_SAMPLER_STRATIFIED) {
				settings.aa_base_x = (uint32_t)get_number(g, "bins", std::max(1u, count)); // StratifiedSampler.cpp:55
			}
		} else if (slot == "lens") {
			settings.lens_samples = count;
		} else if (slot == "time" || slot == "t") {
			settings.time_samples = count;
		} else if (slot == "spectral" || slot == "spectrum" || slot == "s") {
			settings.spectral_samples = count;
			if (const Value* r = g.get("range"))
				if (r->type == Value::GROUP && r->g->anonymous_count() == 2 && r->g->all_numbers()) {
					settings.spectral_start = (float)std::min(r->g->at(0).number(), r->g->at(1).number());
					settings.spectral_end	= (float)std::max(r->g->at(0).number(), r->g->at(1).number());
				}
		} else {
			fail(PRGPU_EINVAL, where(g) + ": unknown sampler slot '" + slot + "'");
		}
	}
	void add_filter(const Group& g) // SceneLoader.cpp:260-306; plugin default radius 3
	{
		const std::string type = lower(get_string(g, "type", ""));
		if (lower(get_string(g, "slot", "pixel")) != "pixel")
			fail(PRGPU_EINVAL, where(g) + ": unknown filter slot");
		if (type == "block" || type == "blur")
			settings.filter = PRGPU_FILTER_BLOCK;
		else if (type == "tri" || type == "triangle")
			settings.filter = PRGPU_FILTER_TRIANGLE;
		else if (type == "gaussian" || type == "gauss")
			settings.filter = PRGPU_FILTER_GAUSSIAN;
		else if (type == "mitchell" || type == "default")
			settings.filter = PRGPU_FILTER_MITCHELL;
		else if (type == "lanczos" || type == "sinc" || type == "lancz")
			settings.filter = PRGPU_FILTER_LANCZOS;
		else
			fail(PRGPU_EINVAL, where(g) + ": unknown filter type '" + type + "'");
		settings.filter_radius = (uint32_t)std::max(0.0, get_number(g, "radius", 3));
		have_filter			   = true;
	}
	void add_mapper(const Group& g) // SceneLoader.cpp:308-352, spd.cpp:360-383, random.cpp, cie.cpp
	{
		const std::string type = lower(get_string(g, "type", ""));
		if (lower(get_string(g, "purpose", "pixel")) != "pixel") {
			warn(where(g) + ": spectral mapper for a purpose other than 'pixel' ignored (only camera paths are traced)");
			return;
		}
		if (type == "random") {
			settings.mapper = PRGPU_MAPPER_RANDOM;
		} else if (type == "spd" || type == "default") {
			settings.mapper = get_bool(g, "cmis", true) ? PRGPU_MAPPER_SPD_CMIS : PRGPU_MAPPER_SPD_HERO;
			if (g.get("bins") || g.get("weighting") || g.get("complete") || g.get("normalized") || g.get("smooth_iterations"))
				fail(PRGPU_EUNSUPPORTED, where(g) + ": only the default spd histogram (bins, weighting, smoothing) is implemented");
		} else if (type == "cie" || type == "cie_y" || type == "visible" || type == "visible_y") { // cie.cpp:107-120
			const bool only_y = type == "cie_y" || type == "visible_y" || get_bool(g, "only_y", false);
			settings.mapper	  = only_y ? PRGPU_MAPPER_CIE_Y : PRGPU_MAPPER_CIE;
		} else if (type == "agh") { // agh.cpp:150-157
			settings.mapper = get_bool(g, "cmis", true) ? PRGPU_MAPPER_AGH_CMIS : PRGPU_MAPPER_AGH_HERO;
		} else {
			fail(PRGPU_EUNSUPPORTED, where(g) + ": spectral mapper '" + type + "' is not supported (spd, random, cie, cie_y, agh are)");
		}
	}
	void add_integrator(const Group& g) // SceneLoader.cpp:354-384, direct.cpp:498-512,545-563
	{
		const std::string type = lower(get_string(g, "type", ""));
		const bool direct	   = type == "direct" || type == "standard" || type == "default";
		if (!direct && !opt.force_direct)
			fail(PRGPU_EUNSUPPORTED, where(g) + ": integrator '" + type + "' is not supported (only direct/standard/default; set force_direct to render with it anyway)");
		if (!direct) {
			warn(where(g) + ": integrator '" + type + "' replaced by 'direct' with default parameters (force_direct)");
			have_integrator = true;
			return;
		}
		settings.max_ray_depth		= (uint32_t)get_number(g, "max_ray_depth", settings.max_ray_depth);
		settings.soft_max_ray_depth = std::min(settings.max_ray_depth, (uint32_t)get_number(g, "soft_max_ray_depth", settings.soft_max_ray_depth));
		const std::string mis		= lower(get_string(g, "mis", "balance"));
		settings.mis				= mis == "power" ? PRGPU_MIS_POWER : PRGPU_MIS_BALANCE;
		settings.emissive_scatter	= get_bool(g, "emissive_scatter", true) ? 1 : 0;
		settings.nee				= get_bool(g, "nee", true) ? 1 : 0; // direct.cpp:512
		if (!get_bool(g, "direct", true))					  // direct.cpp:513: hits of emitters would not count and NEE would lose its MIS partner
			fail(PRGPU_EUNSUPPORTED, where(g) + ": ':direct false' is not supported");
		have_integrator				= true;
	}
	void add_light(const Group& g) // SceneLoader.cpp:558-600, environment.cpp:152-205, distant.cpp:112-121, sun.cpp:250-267, sky.cpp:180-198
	{
		const std::string type = lower(get_string(g, "type", ""));
		prgpu_light l;
		std::memset(&l, 0, sizeof(l));
		transform_of(g, l.transform);
		l.background   = PRGPU_INVALID_ID;
		l.direction[2] = 1.0f;
		string_default = 1.0f; // defaults of :radiance / :background / :irradiance
		if (type == "env" || type == "environment" || type == "background") {
			l.kind					 = PRGPU_LIGHT_ENVIRONMENT;
			const Value *rad = g.get("radiance"), *bg = g.get("background");
			if (rad && bg) {
				l.radiance	 = spectral_node(*rad, g, "radiance");
				l.background = spectral_node(*bg, g, "background");
			} else if (rad) {
				l.radiance = spectral_node(*rad, g, "radiance");
			} else if (bg) {
				l.radiance = spectral_node(*bg, g, "background");
			} else {
				l.radiance = spectrum_const(1.0f);
			}
		} else if (type == "distant" || type == "direction") {
			l.kind	   = PRGPU_LIGHT_DISTANT;
			l.radiance = spectral_param(g, { "irradiance" }, 1.0f);
			float d[3];
			if (get_vec3(g, "direction", d))
				std::memcpy(l.direction, d, sizeof(d));
			if (l.direction[0] == 0 && l.direction[1] == 0 && l.direction[2] == 0)
				fail(PRGPU_EINVAL, where(g) + ": distant light with a zero :direction");
		} else if (type == "sun") { // SunLightFactory::create (sun.cpp:250-267)
			float el, az;
			sun_position(g, el, az);
			const float radius	  = (float)get_number(g, "radius", 1.0);
			const float turbidity = (float)get_number(g, "turbidity", 3.0);
			float scale			  = (float)get_number(g, "power_scale", 1.0);
			const float SUN_VIS_RADIUS = (3.14159265358979323846f / 180.0f) * 0.5358f * 0.5f; // sun.cpp:24
			const bool delta	  = radius <= 1.1920928955078125e-7f;
			if (delta) // SunDeltaLight ctor (sun.cpp:164-170): the solid angle of the visible disc instead of a cone
				scale *= 1.0f;
			else
				scale /= (radius * radius); // SunLight ctor (sun.cpp:42)
			const float theta = 0.5f * 3.14159265358979323846f - el; // ElevationAzimuth::theta
			float values[64];
			const float start = 360.0f, end = 760.0f, delta_nm = (end - start) / (64 - 1);
			const float solid_angle = 2 * 3.14159265358979323846f * (1 - std::cos(SUN_VIS_RADIUS));
			for (int i = 0; i < 64; ++i) {
				const float r = prgpu_sun_radiance(start + i * delta_nm, theta, turbidity);
				values[i]	  = delta ? r * solid_angle * scale : r * scale;
			}
			l.radiance = spectrum_table(start, end, values, 64);
			// ElevationAzimuth::toDirection = Spherical::cartesian(theta, phi) (Spherical.h:36-48)
			l.direction[0] = std::sin(theta) * std::cos(az);
			l.direction[1] = std::sin(theta) * std::sin(az);
			l.direction[2] = std::cos(theta);
			if (delta) {
				l.kind	= PRGPU_LIGHT_DISTANT;
				l.flags = PRGPU_LIGHTF_SUN_DELTA;
			} else {
				l.kind		= PRGPU_LIGHT_SUN;
				l.cos_theta = std::cos(SUN_VIS_RADIUS * radius);
			}
		} else if (type == "sky") { // SkyLightFactory::create (sky.cpp:180-198); the table is the host's SkyModel
			const std::string name = get_string(g, "name", "__unknown");
			const prgpu_prc_sky* sky = nullptr;
			for (uint32_t i = 0; i < opt.n_skies && opt.skies; ++i)
				if (!opt.skies[i].light_name || name == opt.skies[i].light_name) {
					sky = &opt.skies[i];
					break;
				}
			const uint32_t azc = (uint32_t)get_number(g, "azimuth_resolution", 512), elc = (uint32_t)get_number(g, "elevation_resolution", 256);
			if (azc == 0 || elc == 0 || uint64_t(azc) * elc > (1ull << 26))
				fail(PRGPU_EINVAL, where(g) + ": sky resolution out of range");
			// SkyLightFactory::create (sky.cpp:180-198) -> SkyModel (SkyModel.cpp:15-56): ground albedo (default 0.15) at the eleven band
			// wavelengths, the sun position, the turbidity (default 3)
			prgpu_sky_params sp;
			std::memset(&sp, 0, sizeof(sp));
			sun_position(g, sp.sun_elevation, sp.sun_azimuth);
			sp.turbidity = (float)get_number(g, "turbidity", 3.0);
			{ // the albedo is only evaluated here: nodes its expression created do not stay in the scene
				const size_t n_spectra = out.spectra.size(), n_tables = out.tables.size();
				const uint32_t alb	   = spectral_param(g, { "albedo" }, 0.15f);
				for (int k = 0; k < PRGPU_SKY_BANDS; ++k)
					sp.albedo[k] = eval_spectrum_at(alb, 320.0f + k * 40.0f);
				out.spectra.resize(n_spectra);
				out.tables.resize(n_tables);
			}
			std::vector<float> own;
			const float* table = sky ? sky->table : nullptr;
			if (table) {
				if (azc != sky->azimuth_count || elc != sky->elevation_count)
					fail(PRGPU_EINVAL, where(g) + ": the supplied sky table is " + std::to_string(sky->azimuth_count) + " x " + std::to_string(sky->elevation_count)
											+ " but the light asks for " + std::to_string(azc) + " x " + std::to_string(elc));
			} else { // no table from the host: build it here
				own.resize(size_t(azc) * elc * PRGPU_SKY_BANDS);
				if (prgpu_sky_table(sp.sun_elevation, sp.sun_azimuth, sp.turbidity, sp.albedo, azc, elc, own.data()) != PRGPU_OK)
					fail(PRGPU_EINVAL, where(g) + ": the sky model covers turbidities 1 ... 10");
				table = own.data();
			}
			if (out.sky_params.size() < out.lights.size() + 1)
				out.sky_params.resize(out.lights.size() + 1);
			out.sky_params[out.lights.size()] = sp;
			l.kind			  = PRGPU_LIGHT_SKY;
			l.radiance		  = PRGPU_INVALID_ID;
			l.flags			  = (get_bool(g, "extend", true) ? PRGPU_SKYF_EXTEND : 0u) | (get_bool(g, "compensation", false) ? PRGPU_SKYF_COMPENSATION : 0u);
			l.table_offset	  = (uint32_t)out.tables.size();
			l.azimuth_count	  = azc;
			l.elevation_count = elc;
			out.tables.insert(out.tables.end(), table, table + size_t(azc) * elc * PRGPU_SKY_BANDS);
		} else if (type == "uniform_sky" || type == "cloudy_sky") { // CIESkyLightFactory::create, cie_sky.cpp:134-160
			l.kind	= PRGPU_LIGHT_CIE_SKY;
			l.flags = type == "cloudy_sky" ? PRGPU_SKYF_CLOUDY : 0u;
			const Value* zen = g.get("zenith");
			string_default	 = 1.0f;
			l.radiance		 = zen ? spectral_node(*zen, g, "zenith") : spectrum_const(1.0f);
			if (const Value* gt = g.get("ground_tint")) {
				string_default = 0.0f;
				l.background   = spectral_node(*gt, g, "ground_tint");
			}
			string_default		= 1.0f;
			l.ground_brightness = (float)get_number(g, "ground_brightness", 0.2);
		} else {
			fail(PRGPU_EUNSUPPORTED, where(g) + ": light type '" + type + "' is not supported (env/environment/background, distant/direction, sun, sky, uniform_sky and cloudy_sky are)");
		}
		out.lights.push_back(l);
	}
	// OutputSpecification::parse (src/loader/output/io/OutputSpecification.cpp:254-365): the channels of one output file
	void add_output(const Group& g)
	{
		const Value* nm = g.get("name");
		if (!nm || nm->type != Value::STRING) {
			warn(where(g) + ": no name given for output (ignored, as in the reference)");
			return;
		}
		const uint32_t file = (uint32_t)out.output_names.size();
		out.output_names.push_back(nm->s);
		struct Var { const char* str; uint32_t kind, variable; };
		static const Var vars[] = {
			{ "color", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_OUTPUT }, { "spectral", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_OUTPUT },
			{ "output", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_OUTPUT }, { "rgb", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_OUTPUT },
			{ "online_mean", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_ONLINE_MEAN }, { "variance", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_ONLINE_VARIANCE },
			{ "online_variance", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_ONLINE_VARIANCE }, { "var", PRGPU_CHANNEL_SPECTRAL, PRGPU_SPECTRAL_ONLINE_VARIANCE },
			{ "entity_id", PRGPU_CHANNEL_1D, PRGPU_AOV_ENTITY_ID }, { "entity", PRGPU_CHANNEL_1D, PRGPU_AOV_ENTITY_ID }, { "id", PRGPU_CHANNEL_1D, PRGPU_AOV_ENTITY_ID },
			{ "material_id", PRGPU_CHANNEL_1D, PRGPU_AOV_MATERIAL_ID }, { "material", PRGPU_CHANNEL_1D, PRGPU_AOV_MATERIAL_ID }, { "mat", PRGPU_CHANNEL_1D, PRGPU_AOV_MATERIAL_ID },
			{ "emission_id", PRGPU_CHANNEL_1D, PRGPU_AOV_EMISSION_ID }, { "emission", PRGPU_CHANNEL_1D, PRGPU_AOV_EMISSION_ID },
			{ "depth", PRGPU_CHANNEL_1D, PRGPU_AOV_DEPTH }, { "d", PRGPU_CHANNEL_1D, PRGPU_AOV_DEPTH },
			{ "sample_count", PRGPU_CHANNEL_COUNTER, PRGPU_COUNTER_SAMPLES }, { "samples", PRGPU_CHANNEL_COUNTER, PRGPU_COUNTER_SAMPLES }, { "s", PRGPU_CHANNEL_COUNTER, PRGPU_COUNTER_SAMPLES },
			{ "feedback", PRGPU_CHANNEL_COUNTER, PRGPU_COUNTER_FEEDBACK }, { "f", PRGPU_CHANNEL_COUNTER, PRGPU_COUNTER_FEEDBACK }, { "error", PRGPU_CHANNEL_COUNTER, PRGPU_COUNTER_FEEDBACK },
			{ "position", PRGPU_CHANNEL_3D, PRGPU_AOV_POSITION }, { "pos", PRGPU_CHANNEL_3D, PRGPU_AOV_POSITION }, { "p", PRGPU_CHANNEL_3D, PRGPU_AOV_POSITION },
			{ "normal", PRGPU_CHANNEL_3D, PRGPU_AOV_NORMAL }, { "norm", PRGPU_CHANNEL_3D, PRGPU_AOV_NORMAL }, { "n", PRGPU_CHANNEL_3D, PRGPU_AOV_NORMAL },
			{ "normal_geometric", PRGPU_CHANNEL_3D, PRGPU_AOV_NORMAL_G }, { "ng", PRGPU_CHANNEL_3D, PRGPU_AOV_NORMAL_G },
			{ "tangent", PRGPU_CHANNEL_3D, PRGPU_AOV_TANGENT }, { "tan", PRGPU_CHANNEL_3D, PRGPU_AOV_TANGENT }, { "nx", PRGPU_CHANNEL_3D, PRGPU_AOV_TANGENT },
			{ "bitangent", PRGPU_CHANNEL_3D, PRGPU_AOV_BITANGENT }, { "binormal", PRGPU_CHANNEL_3D, PRGPU_AOV_BITANGENT }, { "bi", PRGPU_CHANNEL_3D, PRGPU_AOV_BITANGENT },
			{ "ny", PRGPU_CHANNEL_3D, PRGPU_AOV_BITANGENT }, { "view", PRGPU_CHANNEL_3D, PRGPU_AOV_VIEW }, { "v", PRGPU_CHANNEL_3D, PRGPU_AOV_VIEW },
		};
		// variableToString returns the FIRST spelling of a variable (OutputSpecification.cpp:196-248): the channel's name in the file
		auto canonical = [&](uint32_t kind, uint32_t variable) {
			for (const Var& v : vars)
				if (v.kind == kind && v.variable == variable)
					return std::string(v.str);
			return std::string();
		};
		for (const auto& e : g.entries) {
			if (!e.key.empty() || e.value.type != Value::GROUP)
				continue;
			const Group& c = *e.value.g;
			if (c.id == "custom_channel") {
				warn(where(c) + ": custom output channels are not provided (skipped)");
				continue;
			}
			if (c.id != "channel")
				continue;
			const Value* ty = c.get("type");
			if (!ty || ty->type != Value::STRING)
				continue;
			const std::string type = lower(ty->s);
			std::string lpe;
			if (const Value* l = c.get("lpe"))
				if (l->type == Value::STRING && !l->s.empty()) {
					std::vector<uint8_t> next, accepting;
					std::string lerr;
					if (prgpu_host::compile_lpe(l->s, next, accepting, lerr) != PRGPU_OK || l->s.size() >= sizeof(prgpu_output_channel::lpe)) {
						warn(where(c) + ": invalid or unsupported light path expression '" + l->s + "' (" + lerr + "); the channel is written without it, as in the reference"); // OutputSpecification.cpp:299-302
					} else {
						lpe = l->s;
					}
				}
			if (type == "texture" || type == "uvw" || type == "uv" || type == "tex" || type == "displace_id" || type == "displace") {
				warn(where(c) + ": the '" + type + "' AOV is not provided (skipped)");
				continue;
			}
			const Var* found = nullptr;
			for (const Var& v : vars)
				if (type == v.str) {
					found = &v;
					break;
				}
			if (!found) {
				warn(where(c) + ": unknown channel type '" + type + "' (skipped, as in the reference)");
				continue;
			}
			prgpu_output_channel ch;
			std::memset(&ch, 0, sizeof(ch));
			ch.file		= file;
			ch.kind		= found->kind;
			ch.variable = found->variable;
			ch.tone		= PRGPU_TONE_SRGB;
			const std::string color = lower(get_string(c, "color", ""));
			if (color == "xyz")
				ch.tone = PRGPU_TONE_XYZ;
			else if (color == "norm_xyz")
				ch.tone = PRGPU_TONE_XYZ_NORM;
			else if (color == "lum" || color == "luminance" || color == "gray")
				ch.tone = PRGPU_TONE_LUMINANCE;
			// the colour channel keeps an empty name (R, G, B); every other channel is named after its variable
			std::string name = (found->kind == PRGPU_CHANNEL_SPECTRAL && found->variable == PRGPU_SPECTRAL_OUTPUT) ? std::string() : canonical(found->kind, found->variable);
			if (!lpe.empty()) {
				// colour channels (fragments whose path matches, LocalFrameOutputDevice.cpp:99-113) and shading-point channels (entries
				// whose path matches, :230-249,285-301); raw variance planes and counters have no such variant there
				if (!((found->kind == PRGPU_CHANNEL_SPECTRAL && found->variable == PRGPU_SPECTRAL_OUTPUT) || found->kind == PRGPU_CHANNEL_3D || found->kind == PRGPU_CHANNEL_1D)) {
					warn(where(c) + ": light path expressions are provided for colour and shading-point channels only (channel skipped)");
					continue;
				}
				name += "[" + lpe + "]"; // OutputSpecification.cpp:323-324,335-336,349-350
				std::strncpy(ch.lpe, lpe.c_str(), sizeof(ch.lpe) - 1);
			}
			if (name.size() >= sizeof(ch.name)) {
				warn(where(c) + ": channel name too long (skipped)");
				continue;
			}
			std::strncpy(ch.name, name.c_str(), sizeof(ch.name) - 1);
			out.outputs.push_back(ch);
		}
	}
	// computeSunEA(ParameterGroup) (skysun/SunLocation.cpp:107-130): :direction | :theta :phi | :elevation :azimuth | date, time, location
	void sun_position(const Group& g, float& elevation, float& azimuth)
	{
		const float PI = 3.14159265358979323846f;
		auto from_theta_phi = [&](float theta, float phi) { // ElevationAzimuth::fromThetaPhi
			elevation = 0.5f * PI - theta;
			azimuth	  = phi;
			if (azimuth < 0)
				azimuth += 2 * PI;
		};
		float d[3];
		if (g.get("direction")) {
			if (!get_vec3(g, "direction", d)) {
				d[0] = d[1] = 0;
				d[2]		= 1;
			}
			const float x = (d[0] == 0 && d[1] == 0) ? 1e-5f : d[0]; // Spherical::from_direction (Spherical.h:8-15)
			float phi	  = std::atan2(d[1], x);
			phi			  = phi < 0 ? phi + 2 * PI : phi;
			from_theta_phi(std::acos(d[2]), phi);
		} else if (g.get("theta")) {
			from_theta_phi((float)get_number(g, "theta", 0), (float)get_number(g, "phi", 0));
		} else if (g.get("elevation")) {
			elevation = (float)get_number(g, "elevation", 0);
			azimuth	  = (float)get_number(g, "azimuth", 0);
		} else {
			prgpu_sun_position((int)get_number(g, "year", 2020), (int)get_number(g, "month", 5), (int)get_number(g, "day", 6), (int)get_number(g, "hour", 12),
							   (int)get_number(g, "minute", 0), (float)get_number(g, "seconds", 0.0), (float)get_number(g, "latitude", 49.235422),
							   (float)get_number(g, "longitude", 6.9965744), (float)get_number(g, "timezone", 2), &elevation, &azimuth);
		}
	}
	void add_camera(const Group& g) // perspective.cpp:141-165
	{
		const std::string type = lower(get_string(g, "type", "standard"));
		const bool ortho = type == "ortho" || type == "orthographic"; // ortho.cpp:78-92
		const bool spherical = type == "spherical", fisheye = type == "fisheye"; // spherical.cpp:107-111, fisheye.cpp:175-179
		if (!ortho && !spherical && !fisheye && type != "standard_camera" && type != "standard" && type != "default" && type != "perspective")
			fail(PRGPU_EUNSUPPORTED, where(g) + ": camera type '" + type + "' is not supported (perspective, orthographic, spherical and fisheye are)");
		prgpu_camera c;
		std::memset(&c, 0, sizeof(c));
		c.kind = ortho ? PRGPU_CAMERA_ORTHO : (spherical ? PRGPU_CAMERA_SPHERICAL : (fisheye ? PRGPU_CAMERA_FISHEYE : PRGPU_CAMERA_PERSPECTIVE));
		if (spherical) { // spherical.cpp:96-100
			c.theta_start = (float)get_number(g, "theta_start", 0.0);
			c.theta_end	  = (float)get_number(g, "theta_end", 1.57079632679489661923f);
			c.phi_start	  = (float)get_number(g, "phi_start", -3.14159265358979323846f);
			c.phi_end	  = (float)get_number(g, "phi_end", 3.14159265358979323846f);
		}
		if (fisheye) { // fisheye.cpp:137-173
			const std::string map = lower(get_string(g, "map", "circular"));
			c.fisheye_map		  = map == "cropped" ? PRGPU_FISHEYE_CROPPED : (map == "full" ? PRGPU_FISHEYE_FULL : PRGPU_FISHEYE_CIRCULAR);
			c.fov				  = (float)get_number(g, "fov", 180.0f * (3.14159265358979323846f / 180.0f));
			c.clip_range		  = get_bool(g, "clip_range", true) ? 1u : 0u;
		}
		transform_of(g, c.transform);
		c.width			  = (float)get_number(g, "width", 1);
		c.height		  = (float)get_number(g, "height", 1);
		c.near_t		  = (float)get_number(g, "near", 0.000001); // NEAR_DEFAULT / FAR_DEFAULT, perspective.cpp:14-15
		c.far_t			  = (float)get_number(g, "far", INFINITY);
		c.fstop			  = (float)get_number(g, "fstop", 0);
		c.aperture_radius = (float)get_number(g, "aperture_radius", 0.05);
		const float dd[3] = { 0, 1, 0 }, dr[3] = { 1, 0, 0 }, du[3] = { 0, 0, 1 }; // ICamera::DefaultDirection/Right/Up, core/camera/ICamera.cpp:5-7
		if (!get_vec3(g, "local_direction", c.local_direction))
			std::memcpy(c.local_direction, dd, sizeof(dd));
		if (!get_vec3(g, "local_right", c.local_right))
			std::memcpy(c.local_right, dr, sizeof(dr));
		if (!get_vec3(g, "local_up", c.local_up))
			std::memcpy(c.local_up, du, sizeof(du));
		const std::string name = get_string(g, "name", "__unnamed__");
		cameras[name]		   = c;
		if (first_camera.empty())
			first_camera = name;
	}
	void add_material(const Group& g) // lambert.cpp:90-117
	{
		const std::string type = lower(get_string(g, "type", ""));
		const std::string name = get_string(g, "name", "");
		if (name.empty())
			fail(PRGPU_EINVAL, where(g) + ": material without a name");
		prgpu_material m;
		std::memset(&m, 0, sizeof(m));
		const bool has_roughness = g.get("roughness") || g.get("roughness_x") || g.get("roughness_y");
		auto roughness			 = [&](prgpu_material& mm) { // roughconductor.cpp:158-177 / roughdielectric.cpp:282-310 (scalar nodes: constants only)
			auto scalar = [&](const char* key) -> float {
				const Value* v = g.get(key);
				if (!v)
					return 0.0f; // lookupScalarNode(key, 0)
				if (!v->is_number())
					fail(PRGPU_EUNSUPPORTED, where(g) + ": :" + key + " must be a number (textured roughness is not supported)");
				return (float)v->number();
			};
			mm.roughness_x = g.get("roughness_x") ? scalar("roughness_x") : scalar("roughness");
			if (g.get("roughness_y")) { // its own node: the anisotropic closure, whatever the value
				mm.roughness_y = scalar("roughness_y");
				mm.flags |= PRGPU_MATF_ANISOTROPIC;
			} else {
				mm.roughness_y = mm.roughness_x;
			}
			if (!get_bool(g, "vndf", true))
				mm.flags |= PRGPU_MATF_NO_VNDF;
		};
		const bool rough_glass = type == "roughglass" || type == "roughdielectric" || type == "rough_glass" || type == "rough_dielectric";
		const bool rough_metal = type == "roughconductor" || type == "roughmirror" || type == "roughmetal";
		if (rough_glass || ((type == "glass" || type == "dielectric") && has_roughness)) { // roughdielectric.cpp:282-365, dielectric.cpp:172-176
			m.kind		   = PRGPU_MAT_ROUGH_DIELECTRIC;
			m.ior		   = spectral_param(g, { "eta", "index", "ior" }, 1.55f);
			m.albedo	   = spectral_param(g, { "specularity" }, 1.0f);
			m.transmission = g.get("transmission") ? spectral_node(*g.get("transmission"), g, "transmission") : PRGPU_INVALID_ID;
			roughness(m);
		} else if (rough_metal || ((type == "conductor" || type == "metal") && has_roughness)) { // roughconductor.cpp:158-216, conductor.cpp:102-106
			m.kind		   = PRGPU_MAT_ROUGH_CONDUCTOR;
			m.ior		   = spectral_param(g, { "eta", "index", "ior" }, 1.2f);
			m.k			   = spectral_param(g, { "k", "kappa" }, 2.605f);
			m.albedo	   = spectral_param(g, { "specularity" }, 1.0f);
			m.transmission = PRGPU_INVALID_ID;
			roughness(m);
		} else if (type == "glass" || type == "dielectric") { // dielectric.cpp:150-197
			m.kind		   = PRGPU_MAT_DIELECTRIC;
			m.ior		   = spectral_param(g, { "index", "eta", "ior" }, 1.55f);
			m.albedo	   = spectral_param(g, { "specularity" }, 1.0f);
			m.transmission = g.get("transmission") ? spectral_node(*g.get("transmission"), g, "transmission") : PRGPU_INVALID_ID;
			m.thin		   = get_bool(g, "thin", false) ? 1 : 0;
		} else if (type == "conductor" || type == "metal") { // conductor.cpp:95-125
			m.kind		   = PRGPU_MAT_CONDUCTOR;
			m.ior		   = spectral_param(g, { "eta", "index", "ior" }, 1.2f);
			m.k			   = spectral_param(g, { "k", "kappa" }, 2.605f);
			m.albedo	   = spectral_param(g, { "specularity" }, 1.0f);
			m.transmission = PRGPU_INVALID_ID;
		} else if (type == "principled") { // principled.cpp:634-687 (scalar nodes: constants only)
			m.kind		   = PRGPU_MAT_PRINCIPLED;
			m.albedo	   = spectral_param(g, { "base_color", "base" }, 0.8f);
			m.ior		   = spectral_param(g, { "ior", "eta", "index" }, 1.55f);
			m.transmission = PRGPU_INVALID_ID;
			m.thin		   = get_bool(g, "thin", false) ? 1 : 0;
			auto scalar	   = [&](std::initializer_list<const char*> keys, float def) -> float {
				   for (const char* key : keys)
					   if (const Value* v = g.get(key)) {
						   if (!v->is_number())
							   fail(PRGPU_EUNSUPPORTED, where(g) + ": :" + key + " must be a number (textured principled parameters are not supported)");
						   return (float)v->number();
					   }
				   return def;
			};
			m.roughness_x = m.roughness_y						= scalar({ "roughness" }, 0.5f);
			m.principled[PRGPU_PRINCIPLED_DIFFUSE_TRANSMISSION]	= scalar({ "diffuse_transmission", "diff_trans" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_SPECULAR_TRANSMISSION] = scalar({ "specular_transmission", "spec_trans" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_SPECULAR_TINT]		= scalar({ "specular_tint" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_ANISOTROPIC]			= scalar({ "anisotropic" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_FLATNESS]				= scalar({ "flatness", "subsurface" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_METALLIC]				= scalar({ "metallic" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_SHEEN]				= scalar({ "sheen" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_SHEEN_TINT]			= scalar({ "sheen_tint" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_CLEARCOAT]			= scalar({ "clearcoat" }, 0.0f);
			m.principled[PRGPU_PRINCIPLED_CLEARCOAT_GLOSS]		= scalar({ "clearcoat_gloss" }, 0.0f);
			if (g.get("specular_transmission") || g.get("spec_trans") || g.get("diffuse_transmission") || g.get("diff_trans")) // :657-666
				m.flags |= PRGPU_MATF_HAS_TRANSMISSION;
			if (!get_bool(g, "vndf", true)) // PrincipledMaterialPlugin::create (principled.cpp:677-685)
				m.flags |= PRGPU_MATF_NO_VNDF;
		} else if (type == "mirror" || type == "reflection") { // mirror.cpp:79-100
			m.kind		   = PRGPU_MAT_MIRROR;
			m.albedo	   = spectral_param(g, { "specularity" }, 1.0f);
			m.transmission = PRGPU_INVALID_ID;
		} else if (type == "diffuse" || type == "lambert") {
			m.kind		= PRGPU_MAT_LAMBERT;
			m.albedo	= spectral_param(g, { "albedo", "base", "diffuse" }, 1.0f);
			m.two_sided = get_bool(g, "two_sided", true) ? 1 : 0;
		} else {
			fail(PRGPU_EUNSUPPORTED, where(g) + ": material type '" + type + "' is not supported (diffuse/lambert, glass/dielectric, conductor/metal, their rough variants and principled are)");
		}
		material_ids[name] = (uint32_t)out.materials.size();
		out.materials.push_back(m);
	}
	void add_emission(const Group& g) // diffuse.cpp:55-80
	{
		const std::string type = lower(get_string(g, "type", ""));
		const std::string name = get_string(g, "name", "");
		if (name.empty())
			fail(PRGPU_EINVAL, where(g) + ": emission without a name");
		if (type != "diffuse" && type != "standard" && type != "default")
			fail(PRGPU_EUNSUPPORTED, where(g) + ": emission type '" + type + "' is not supported (diffuse only)");
		prgpu_emission e;
		e.kind	   = PRGPU_EMS_DIFFUSE;
		e.radiance = spectral_param(g, { "radiance" }, 1.0f);
		emission_ids[name] = (uint32_t)out.emissions.size();
		out.emissions.push_back(e);
	}
	void add_mesh(const Group& g) // MeshParser.cpp:134-255
	{
		const std::string name = get_string(g, "name", "");
		if (name.empty())
			fail(PRGPU_EINVAL, where(g) + ": mesh without a name");
		Mesh m;
		auto load_attr = [&](const Group& a, std::vector<float>& dst) {
			for (size_t j = 0; j < a.anonymous_count(); ++j) {
				const Value& v = a.at(j);
				const size_t n = v.type == Value::GROUP ? v.g->anonymous_count() : 0;
				if (v.type != Value::GROUP || !v.g->all_numbers() || (n != 2 && n != 3))
					fail(PRGPU_EINVAL, where(a) + " of mesh '" + name + "': attribute entry " + std::to_string(j) + " is not a vector");
				for (size_t k = 0; k < 3; ++k)
					dst.push_back(k < n ? (float)v.g->at(k).number() : 0.0f);
			}
		};
		for (const auto& e : g.entries) {
			if (!e.key.empty())
				continue;
			if (e.value.type != Value::GROUP)
				fail(PRGPU_EINVAL, where(g) + ": invalid entry in mesh description");
			const Group& b = *e.value.g;
			if (b.id == "attribute") {
				const std::string t = get_string(b, "type", "");
				if (t == "p")
					load_attr(b, m.p);
				else if (t == "n")
					load_attr(b, m.n);
				else if (t == "t" || t == "uv") { // texture coordinates: two components per vertex
					std::vector<float> tmp;
					load_attr(b, tmp);
					for (size_t j = 0; j + 2 < tmp.size() + 1 && j < tmp.size(); j += 3) {
						m.uv.push_back(tmp[j]);
						m.uv.push_back(tmp[j + 1]);
					}
				} else if (t == "w" || t == "dp" || t == "u")
					; // weights, velocities, user attributes: not evaluated on this path
				else
					fail(PRGPU_EINVAL, where(b) + ": unknown mesh attribute '" + t + "'");
			} else if (b.id == "faces") {
				for (size_t j = 0; j < b.anonymous_count(); ++j) {
					const Value& v = b.at(j);
					if (v.type != Value::GROUP || !v.g->is_array)
						fail(PRGPU_EINVAL, where(b) + " of mesh '" + name + "': face " + std::to_string(j) + " is not an index array");
					const size_t n = v.g->anonymous_count();
					if (n != 3 && n != 4)
						fail(PRGPU_EINVAL, where(b) + " of mesh '" + name + "': only triangle or quad faces are supported");
					std::vector<uint32_t> f;
					for (size_t k = 0; k < n; ++k) {
						if (v.g->at(k).type != Value::INT || v.g->at(k).i < 0)
							fail(PRGPU_EINVAL, where(b) + " of mesh '" + name + "': face index is not a non-negative integer");
						f.push_back((uint32_t)v.g->at(k).i);
					}
					m.faces.push_back(f);
				}
			} else if (b.id == "materials") {
				for (size_t j = 0; j < b.anonymous_count(); ++j) {
					if (b.at(j).type != Value::INT || b.at(j).i < 0)
						fail(PRGPU_EINVAL, where(b) + " of mesh '" + name + "': material slot is not a non-negative integer");
					m.slots.push_back((uint32_t)b.at(j).i);
				}
			} else if (b.id == "normal_faces" || b.id == "texture_faces" || b.id == "weight_faces" || b.id == "velocity_faces") {
				if (b.id == "normal_faces")
					fail(PRGPU_EUNSUPPORTED, where(b) + " of mesh '" + name + "': separate normal indices are not supported");
			}
		}
		const size_t nv = m.p.size() / 3;
		if (nv == 0 || m.faces.empty())
			fail(PRGPU_EINVAL, "mesh '" + name + "' has no vertices or faces");
		if (!m.n.empty() && m.n.size() != m.p.size())
			fail(PRGPU_EINVAL, "mesh '" + name + "': normal count differs from vertex count");
		if (!m.uv.empty() && m.uv.size() / 2 != nv)
			fail(PRGPU_EINVAL, "mesh '" + name + "': texture coordinate count differs from vertex count");
		if (!m.uv.empty())
			for (const auto& f : m.faces)
				if (f.size() == 4) // Face::tangentFromUV / interpolateUVs work on the whole quad (Face.h:39-45,80-98), not on its two triangles
					fail(PRGPU_EUNSUPPORTED, "mesh '" + name + "': quads with texture coordinates are not supported (triangulate the mesh)");
		if (!m.slots.empty() && m.slots.size() != m.faces.size())
			fail(PRGPU_EINVAL, "mesh '" + name + "': material slot count differs from face count");
		for (const auto& f : m.faces)
			for (uint32_t i : f)
				if (i >= nv)
					fail(PRGPU_EINVAL, "mesh '" + name + "': face index out of range");
		meshes[name] = std::move(m);
	}
	// Entity visibility flags are parsed by the reference (SceneLoader.cpp:452-497) and stored, but no tracing code ever reads them:
	// every Embree ray carries MASK_ALL (Scene.cpp:135,166).  Accept them and say so.
	void note_visibility_flags(const Group& g)
	{
		for (const char* flag : { "camera_visible", "light_visible", "bounce_visible", "shadow_visible" })
			if (!get_bool(g, flag, true))
				warn(where(g) + ": visibility flag :" + flag + " false has no effect (the reference stores it and traces with MASK_ALL)");
	}
	void add_plane(const Group& g, const std::string& name) // plane.cpp:241-258
	{
		note_visibility_flags(g);
		float xa[3] = { 1, 0, 0 }, ya[3] = { 0, 1, 0 };
		if (!get_vec3(g, "x_axis", xa))
			get_vec3(g, "axis_x", xa);
		if (!get_vec3(g, "y_axis", ya))
			get_vec3(g, "axis_y", ya);
		const float w = (float)get_number(g, "width", 1), h = (float)get_number(g, "height", 1);
		float x[3], y[3], p[3];
		for (int k = 0; k < 3; ++k) {
			x[k] = w * xa[k];
			y[k] = h * ya[k];
		}
		const bool centering = get_bool(g, "centering", false);
		for (int k = 0; k < 3; ++k)
			p[k] = centering ? -0.5f * x[k] - 0.5f * y[k] : 0.0f; // PlaneEntity::centerOn
		const Value* mv = g.get("material");
		uint32_t mat	= PRGPU_INVALID_ID;
		if (mv && mv->type == Value::STRING) {
			const auto it = material_ids.find(mv->s);
			mat			  = it == material_ids.end() ? PRGPU_INVALID_ID : it->second;
		}
		prgpu_entity e;
		std::memset(&e, 0, sizeof(e));
		e.first_tri = (uint32_t)(out.indices.size() / 3);
		e.n_tris	= 2;
		e.emission	= PRGPU_INVALID_ID;
		{
			const Value* ev = g.get("emission");
			const auto it	= ev && ev->type == Value::STRING ? emission_ids.find(ev->s) : emission_ids.end();
			if (ev && it == emission_ids.end())
				fail(PRGPU_EINVAL, where(g) + ": entity '" + name + "' refers to an unknown emission");
			if (ev)
				e.emission = it->second;
		}
		e.kind		= PRGPU_ENTITY_PLANE;
		transform_of(g, e.transform);
		const uint32_t base = (uint32_t)(out.positions.size() / 3);
		const float v[4][3] = { { p[0], p[1], p[2] },
								{ p[0] + y[0], p[1] + y[1], p[2] + y[2] },
								{ (p[0] + y[0]) + x[0], (p[1] + y[1]) + x[1], (p[2] + y[2]) + x[2] },
								{ p[0] + x[0], p[1] + x[1], p[2] + x[2] } }; // plane.cpp:81-84
		for (const auto& q : v)
			out.positions.insert(out.positions.end(), q, q + 3);
		const uint32_t idx[6] = { 0, 1, 3, 2, 3, 1 }; // Embree quad
		for (uint32_t i : idx)
			out.indices.push_back(base + i);
		out.tri_material.push_back(mat);
		out.tri_material.push_back(mat);
		out.entities.push_back(e);
		(void)name;
	}
	void add_entity(const Group& g) // SceneLoader.cpp:446-520, mesh.cpp:260-300
	{
		const std::string type = lower(get_string(g, "type", ""));
		const std::string name = get_string(g, "name", "__unnamed__");
		if (type == "plane") {
			add_plane(g, name);
			return;
		}
		if (type == "sphere") { // sphere.cpp:157-168
			note_visibility_flags(g);
			prgpu_entity e;
			std::memset(&e, 0, sizeof(e));
			e.first_tri = (uint32_t)(out.indices.size() / 3);
			e.n_tris	= 1;
			e.emission	= PRGPU_INVALID_ID;
			{
				const Value* ev = g.get("emission");
				const auto it	= ev && ev->type == Value::STRING ? emission_ids.find(ev->s) : emission_ids.end();
				if (ev && it == emission_ids.end())
					fail(PRGPU_EINVAL, where(g) + ": entity '" + name + "' refers to an unknown emission");
				if (ev)
					e.emission = it->second;
			}
			e.kind		= PRGPU_ENTITY_SPHERE;
			e.radius	= (float)get_number(g, "radius", 1.0);
			if (!(e.radius > 0))
				fail(PRGPU_EINVAL, where(g) + ": sphere :radius must be positive");
			transform_of(g, e.transform);
			const Value* mv = g.get("material");
			uint32_t mat	= PRGPU_INVALID_ID;
			if (mv && mv->type == Value::STRING) {
				const auto it = material_ids.find(mv->s);
				mat			  = it == material_ids.end() ? PRGPU_INVALID_ID : it->second;
			}
			const uint32_t base = (uint32_t)(out.positions.size() / 3);
			out.positions.insert(out.positions.end(), 9, 0.0f); // placeholder triangle
			for (uint32_t i = 0; i < 3; ++i)
				out.indices.push_back(base + i);
			out.tri_material.push_back(mat);
			out.entities.push_back(e);
			return;
		}
		if (type == "quadric" || type == "cone" || type == "cylinder") { // QuadricEntityPlugin::create (quadric.cpp:252-316)
			note_visibility_flags(g);
			prgpu_entity e;
			std::memset(&e, 0, sizeof(e));
			e.first_tri = (uint32_t)(out.indices.size() / 3);
			e.n_tris	= 1;
			e.emission	= PRGPU_INVALID_ID;
			if (g.get("emission"))
				fail(PRGPU_EUNSUPPORTED, where(g) + ": emissive " + type + " entities are not supported (the reference's sampler for them is a stub with pdf 0)");
			float q[16] = { 0 };
			if (type == "quadric") {
				const Value* pv = g.get("parameters");
				std::vector<float> qp;
				if (pv && pv->type == Value::GROUP && pv->g->is_array && pv->g->all_numbers())
					for (size_t i = 0; i < pv->g->anonymous_count(); ++i)
						qp.push_back((float)pv->g->at(i).number());
				if (qp.size() == 3 || qp.size() == 4) {
					q[0] = qp[0], q[1] = qp[1], q[2] = qp[2];
					q[9] = qp.size() == 4 ? qp[3] : 0.0f;
				} else if (qp.size() == 10) {
					std::copy(qp.begin(), qp.end(), q);
				} else {
					fail(PRGPU_EINVAL, where(g) + ": invalid quadric parameters given (3, 4 or 10 numbers)"); // quadric.cpp:308-310
				}
				float lo[3] = { -1, -1, -1 }, hi[3] = { 1, 1, 1 };
				(void)get_vec3(g, "min", lo);
				(void)get_vec3(g, "max", hi);
				for (int k = 0; k < 3; ++k) {
					q[10 + k] = lo[k];
					q[13 + k] = hi[k];
				}
			} else {
				const float radius = (float)get_number(g, "radius", 1.0), height = (float)get_number(g, "height", 1.0);
				const bool center  = get_bool(g, "center_on", true);
				if (!(radius > 0) || !(height > 0))
					fail(PRGPU_EINVAL, where(g) + ": " + type + " :radius and :height must be positive");
				const float a2 = 1 / (radius * radius);
				q[0] = q[1] = a2;
				if (type == "cylinder") {
					q[9] = -1.0f;
				} else {
					q[2] = -(1 / (height * height));
					if (center) {
						q[8] = 1 / height;
						q[9] = -0.25f;
					}
				}
				q[10] = q[11] = -radius;
				q[13] = q[14] = radius;
				q[12] = center ? -height / 2 : 0.0f;
				q[15] = center ? height / 2 : height;
			}
			e.kind	 = PRGPU_ENTITY_QUADRIC;
			e.params = (uint32_t)out.tables.size();
			out.tables.insert(out.tables.end(), q, q + 16);
			transform_of(g, e.transform);
			const Value* mv = g.get("material");
			uint32_t mat	= PRGPU_INVALID_ID;
			if (mv && mv->type == Value::STRING) {
				const auto it = material_ids.find(mv->s);
				mat			  = it == material_ids.end() ? PRGPU_INVALID_ID : it->second;
			}
			const uint32_t base = (uint32_t)(out.positions.size() / 3);
			out.positions.insert(out.positions.end(), 9, 0.0f); // placeholder triangle: one point, never hit
			for (uint32_t i = 0; i < 3; ++i)
				out.indices.push_back(base + i);
			out.tri_material.push_back(mat);
			out.entities.push_back(e);
			return;
		}
		if (type != "mesh")
			fail(PRGPU_EUNSUPPORTED, where(g) + ": entity type '" + type + "' is not supported (mesh, plane, sphere, quadric, cone and cylinder are; tessellate other primitives)");
		note_visibility_flags(g);
		const auto mit = meshes.find(get_string(g, "mesh", ""));
		if (mit == meshes.end())
			fail(PRGPU_EINVAL, where(g) + ": entity '" + name + "' refers to unknown mesh '" + get_string(g, "mesh", "") + "'");
		const Mesh& m = mit->second;
		std::vector<uint32_t> mats; // SceneLoadContext::lookupMaterialIDArray
		const Value* mv = g.get("materials") ? g.get("materials") : g.get("material");
		auto lookup = [&](const std::string& n) {
			const auto it = material_ids.find(n);
			return it == material_ids.end() ? PRGPU_INVALID_ID : it->second; // unknown names give PR_INVALID_ID like the reference
		};
		if (mv && mv->type == Value::STRING)
			mats.push_back(lookup(mv->s));
		else if (mv && mv->type == Value::GROUP && mv->g->is_array)
			for (size_t i = 0; i < mv->g->anonymous_count(); ++i)
				mats.push_back(mv->g->at(i).type == Value::STRING ? lookup(mv->g->at(i).s) : PRGPU_INVALID_ID);
		prgpu_entity e;
		std::memset(&e, 0, sizeof(e));
		e.first_tri	  = (uint32_t)(out.indices.size() / 3);
		e.emission	  = PRGPU_INVALID_ID;
		if (const Value* ev = g.get("emission")) {
			const auto it = ev->type == Value::STRING ? emission_ids.find(ev->s) : emission_ids.end();
			if (it == emission_ids.end())
				fail(PRGPU_EINVAL, where(g) + ": entity '" + name + "' refers to an unknown emission");
			e.emission = it->second;
		}
		e.has_normals = (!m.n.empty() && !get_bool(g, "ignore_normals", false)) ? 1 : 0;
		transform_of(g, e.transform);
		const uint32_t base = (uint32_t)(out.positions.size() / 3);
		out.positions.insert(out.positions.end(), m.p.begin(), m.p.end());
		if (e.has_normals) {
			out.normals.resize(size_t(base) * 3, 0.0f);
			out.normals.insert(out.normals.end(), m.n.begin(), m.n.end());
			any_normals = true;
		}
		e.has_uvs = m.uv.empty() ? 0 : 1;
		if (e.has_uvs) {
			out.uvs.resize(size_t(base) * 2, 0.0f);
			out.uvs.insert(out.uvs.end(), m.uv.begin(), m.uv.end());
			any_uvs = true;
		}
		for (size_t fi = 0; fi < m.faces.size(); ++fi) {
			const auto& f		= m.faces[fi];
			const uint32_t slot = m.slots.empty() ? 0u : m.slots[fi];
			const uint32_t mat	= slot < mats.size() ? mats[slot] : PRGPU_INVALID_ID; // mesh.cpp:227
			auto tri = [&](uint32_t a, uint32_t b, uint32_t c) {
				out.indices.push_back(base + a);
				out.indices.push_back(base + b);
				out.indices.push_back(base + c);
				out.tri_material.push_back(mat);
			};
			if (f.size() == 3) {
				tri(f[0], f[1], f[2]);
			} else { // quads are handed to Embree as RTC_GEOMETRY_TYPE_QUAD: triangles (v0,v1,v3) and (v2,v3,v1)
				tri(f[0], f[1], f[3]);
				tri(f[2], f[3], f[1]);
			}
		}
		e.n_tris = (uint32_t)(out.indices.size() / 3) - e.first_tri;
		out.entities.push_back(e);
	}

	// (embed :loader 'obj' :file 'x.obj' :name 'n' [:flipNormal bool]) -- SceneLoader.cpp:775-812 + archives/WavefrontLoader.cpp:24-170.
	// The reference reads the file with tinyobjloader (un-vendored submodule external/tinyobjloader, PearCoding fork, branch master,
	// no pinned revision; LoadObj(..., triangulate = true)).  Restated here from the OBJ format and tinyobj's documented behaviour:
	// `v`/`vn` records, `f` corners "v", "v/vt", "v//vn", "v/vt/vn" with 1-based or negative (relative) indices, polygons
	// triangulated as a fan around their first corner, all shapes merged into one mesh, materials ignored.  When the normal indices
	// differ from the vertex indices the reference keeps a second index buffer (WavefrontLoader.cpp:55-64,118-128); this backend's
	// meshes have one index per corner, so such a mesh is expanded to one vertex per face corner (same geometry, same normals).
	void add_embed(const Group& g, const std::string& dir)
	{
		const std::string loader = get_string(g, "loader", "obj"); // compared as written (SceneLoader.cpp:799-842)
		if (loader != "obj" && loader != "ply" && loader != "mts")
			fail(PRGPU_EUNSUPPORTED, where(g) + ": embed loader '" + loader + "' is not supported (obj, ply and mts are)");
		std::string path = get_string(g, "file", "");
		if (path.empty())
			fail(PRGPU_EINVAL, where(g) + ": embed without a :file");
		if (path[0] != '/' && !dir.empty())
			path = dir + "/" + path;
		if (loader == "ply") {
			add_embed_ply(g, path);
			return;
		}
		if (loader == "mts") {
			add_embed_mts(g, path);
			return;
		}
		std::ifstream f(path);
		if (!f)
			fail(PRGPU_EINVAL, where(g) + ": cannot open '" + path + "'");
		const bool flip = get_bool(g, "flipNormal", false);
		std::vector<float> v, vn;
		struct Corner {
			int v, n;
		};
		std::vector<std::vector<Corner>> faces;
		std::string name = get_string(g, "name", ""), first_shape, line;
		int line_no = 0;
		while (std::getline(f, line)) {
			++line_no;
			std::istringstream ls(line);
			std::string tag;
			if (!(ls >> tag) || tag[0] == '#')
				continue;
			if (tag == "v" || tag == "vn") {
				float x[3];
				if (!(ls >> x[0] >> x[1] >> x[2]))
					fail(PRGPU_EINVAL, path + ":" + std::to_string(line_no) + ": malformed '" + tag + "' record");
				std::vector<float>& dst = tag == "v" ? v : vn;
				for (float c : x)
					dst.push_back(tag == "vn" && flip ? -c : c);
			} else if (tag == "f") {
				std::vector<Corner> face;
				std::string tok;
				while (ls >> tok) {
					Corner c{ 0, 0 };
					const size_t s1 = tok.find('/');
					c.v				= std::atoi(tok.substr(0, s1).c_str());
					if (s1 != std::string::npos) {
						const size_t s2 = tok.find('/', s1 + 1);
						if (s2 != std::string::npos && s2 + 1 < tok.size())
							c.n = std::atoi(tok.substr(s2 + 1).c_str());
					}
					const int nv = (int)(v.size() / 3), nn = (int)(vn.size() / 3);
					c.v = c.v < 0 ? nv + c.v : c.v - 1; // negative indices are relative to the records read so far
					c.n = c.n < 0 ? nn + c.n : c.n - 1; // -1: no normal given
					if (c.v < 0 || c.v >= nv || c.n >= nn)
						fail(PRGPU_EINVAL, path + ":" + std::to_string(line_no) + ": face index out of range");
					face.push_back(c);
				}
				if (face.size() < 3)
					fail(PRGPU_EINVAL, path + ":" + std::to_string(line_no) + ": face with fewer than three corners");
				faces.push_back(face);
			} else if ((tag == "o" || tag == "g") && first_shape.empty()) {
				ls >> first_shape;
			}
		}
		if (v.empty() || faces.empty())
			fail(PRGPU_EINVAL, path + ": no vertices or faces");
		if (name.empty())
			name = first_shape; // WavefrontLoader.cpp:150-153
		if (name.empty())
			fail(PRGPU_EINVAL, where(g) + ": embedded mesh has no name");
		bool use_normals = !vn.empty(), same_index = true;
		for (const auto& face : faces)
			for (const Corner& c : face) {
				if (c.n < 0)
					use_normals = false;
				if (c.n != c.v)
					same_index = false;
			}
		Mesh m;
		if (!use_normals || (same_index && vn.size() == v.size())) {
			m.p = v;
			if (use_normals)
				m.n = vn;
			for (const auto& face : faces)
				for (size_t k = 1; k + 1 < face.size(); ++k)
					m.faces.push_back({ (uint32_t)face[0].v, (uint32_t)face[k].v, (uint32_t)face[k + 1].v });
		} else { // one vertex per face corner
			for (const auto& face : faces) {
				const uint32_t base = (uint32_t)(m.p.size() / 3);
				for (const Corner& c : face) {
					m.p.insert(m.p.end(), v.begin() + 3 * c.v, v.begin() + 3 * c.v + 3);
					m.n.insert(m.n.end(), vn.begin() + 3 * c.n, vn.begin() + 3 * c.n + 3);
				}
				for (uint32_t k = 1; k + 1 < face.size(); ++k)
					m.faces.push_back({ base, base + k, base + k + 1 });
			}
		}
		meshes[name] = std::move(m);
	}

	// PlyLoader::load (src/loader/archives/PlyLoader.cpp:225-333): ascii / binary little / big endian, float vertex properties
	// x y z [nx ny nz] [u v] (other float properties are skipped by position), faces as `list <count type> <index type> vertex_indices`
	// of triangles and quads; normals are normalised on load (:141-148)
	void add_embed_ply(const Group& g, const std::string& path)
	{
		std::ifstream f(path, std::ios::in | std::ios::binary);
		if (!f)
			fail(PRGPU_EINVAL, where(g) + ": cannot open '" + path + "'");
		std::string magic;
		f >> magic;
		if (magic != "ply")
			fail(PRGPU_EINVAL, path + ": not a ply file");
		std::string method, line;
		int n_vertices = 0, n_faces = 0, elem[8] = { -1, -1, -1, -1, -1, -1, -1, -1 } /* x y z nx ny nz u v */, n_props = 0, ind_elem = -1, face_props = 0;
		static const char* prop_names[8] = { "x", "y", "z", "nx", "ny", "nz", "u", "v" };
		while (std::getline(f, line)) {
			std::istringstream ls(line);
			std::string action;
			ls >> action;
			if (action == "comment")
				continue;
			if (action == "format") {
				ls >> method;
			} else if (action == "element") {
				std::string type;
				ls >> type;
				if (type == "vertex")
					ls >> n_vertices;
				else if (type == "face")
					ls >> n_faces;
			} else if (action == "property") {
				std::string type;
				ls >> type;
				if (type == "float") {
					std::string name;
					ls >> name;
					for (int k = 0; k < 8; ++k)
						if (name == prop_names[k])
							elem[k] = n_props;
					++n_props;
				} else if (type == "list") {
					++face_props;
					std::string count_type, ind_type, name;
					ls >> count_type >> ind_type >> name;
					if (count_type != "uchar" && count_type != "int" && count_type != "uint8" && count_type != "uint") {
						warn(path + ": only 'property list uchar int' is supported");
						continue;
					}
					if (name == "vertex_indices")
						ind_elem = face_props - 1;
				} else {
					warn(path + ": only float or list properties are read; '" + type + "' is skipped as one 4-byte property");
					++n_props;
				}
			} else if (action == "end_header") {
				break;
			}
		}
		if (elem[0] < 0 || elem[1] < 0 || elem[2] < 0 || ind_elem < 0 || n_vertices <= 0 || n_faces <= 0)
			fail(PRGPU_EINVAL, path + ": the ply file does not contain valid mesh data");
		const bool ascii = method == "ascii", swap = method == "binary_big_endian";
		const bool has_n = elem[3] >= 0 && elem[4] >= 0 && elem[5] >= 0, has_uv = elem[6] >= 0 && elem[7] >= 0;
		auto read_f32 = [&]() {
			unsigned char b[4] = { 0, 0, 0, 0 };
			f.read(reinterpret_cast<char*>(b), 4);
			if (swap)
				std::swap(b[0], b[3]), std::swap(b[1], b[2]);
			float v;
			std::memcpy(&v, b, 4);
			return v;
		};
		Mesh m;
		for (int i = 0; i < n_vertices; ++i) {
			float val[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			if (ascii) {
				if (!std::getline(f, line))
					fail(PRGPU_EINVAL, path + ": not enough vertices given");
				std::istringstream ls(line);
				float x;
				for (int e = 0; ls >> x; ++e)
					for (int k = 0; k < 8; ++k)
						if (elem[k] == e)
							val[k] = x;
			} else {
				for (int e = 0; e < n_props; ++e) {
					const float x = read_f32();
					for (int k = 0; k < 8; ++k)
						if (elem[k] == e)
							val[k] = x;
				}
			}
			m.p.insert(m.p.end(), val, val + 3);
			if (has_n) {
				float norm = std::sqrt(val[3] * val[3] + val[4] * val[4] + val[5] * val[5]);
				if (norm <= 1.1920928955078125e-7f)
					norm = 1.0f;
				for (int k = 3; k < 6; ++k)
					m.n.push_back(val[k] / norm);
			}
			if (has_uv)
				m.uv.insert(m.uv.end(), val + 6, val + 8);
		}
		if (!ascii && !f)
			fail(PRGPU_EINVAL, path + ": not enough vertices given");
		for (int i = 0; i < n_faces; ++i) {
			std::vector<uint32_t> face;
			if (ascii) {
				if (!std::getline(f, line))
					fail(PRGPU_EINVAL, path + ": not enough indices given");
				std::istringstream ls(line);
				uint32_t n = 0, ind;
				ls >> n;
				for (uint32_t j = 0; j < n && (ls >> ind); ++j)
					face.push_back(ind);
			} else {
				unsigned char n = 0;
				f.read(reinterpret_cast<char*>(&n), 1);
				for (unsigned j = 0; j < n; ++j) {
					unsigned char b[4] = { 0, 0, 0, 0 };
					f.read(reinterpret_cast<char*>(b), 4);
					if (swap)
						std::swap(b[0], b[3]), std::swap(b[1], b[2]);
					uint32_t ind;
					std::memcpy(&ind, b, 4);
					face.push_back(ind);
				}
				if (!f)
					fail(PRGPU_EINVAL, path + ": not enough indices given");
			}
			if (face.size() != 3 && face.size() != 4)
				fail(PRGPU_EINVAL, path + ": only triangles or quads are allowed in ply files");
			for (uint32_t ind : face)
				if (ind >= (uint32_t)n_vertices)
					fail(PRGPU_EINVAL, path + ": face index out of range");
			m.faces.push_back(face);
		}
		const std::string name = get_string(g, "name", "");
		if (name.empty())
			fail(PRGPU_EINVAL, where(g) + ": embedded mesh has no name");
		meshes[name] = std::move(m);
	}
	// MtsSerializedLoader::load (src/loader/archives/MtsSerializedLoader.cpp:130-328): Mitsuba's .serialized container -- u16 0x041C,
	// u16 version (>= 3), zlib streams per shape, the shape offsets (u64, u32 for version 3) and the shape count at the end of the
	// file; a stream holds flags, [name], vertex and triangle counts, positions, [normals], [uvs], [colours], indices.
	void add_embed_mts(const Group& g, const std::string& path)
	{
		std::ifstream f(path, std::ios::in | std::ios::binary);
		if (!f)
			fail(PRGPU_EINVAL, where(g) + ": cannot open '" + path + "'");
		std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
		auto rd = [&](size_t off, void* dst, size_t n) {
			if (off + n > file.size())
				fail(PRGPU_EINVAL, path + ": truncated Mitsuba serialized file");
			std::memcpy(dst, file.data() + off, n);
		};
		uint16_t ident = 0, version = 0;
		rd(0, &ident, 2);
		if (ident != 0x041C)
			fail(PRGPU_EINVAL, path + ": not a valid Mitsuba serialized file");
		rd(2, &version, 2);
		if (version < 3)
			fail(PRGPU_EINVAL, path + ": insufficient version number " + std::to_string(version) + " < 3");
		const uint32_t shape = (uint32_t)get_number(g, "shape", 0);
		uint32_t count = 0;
		rd(file.size() - 4, &count, 4);
		if (shape >= count)
			fail(PRGPU_EINVAL, path + ": cannot access shape " + std::to_string(shape) + ", the file contains " + std::to_string(count));
		const size_t osz = version >= 4 ? 8 : 4;
		if (uint64_t(count) * osz + 4 > file.size()) // the offset dictionary at the end of the file must fit into it
			fail(PRGPU_EINVAL, path + ": the shape count " + std::to_string(count) + " does not fit the file");
		auto offset_of = [&](uint32_t k) {
			uint64_t o = 0;
			rd(file.size() - 4 - osz * (count - k), &o, osz);
			return o;
		};
		const uint64_t start = offset_of(shape), end = shape == count - 1 ? file.size() - 4 - 0 : offset_of(shape + 1);
		if (start + 4 > end || end > file.size())
			fail(PRGPU_EINVAL, path + ": bad shape offsets");
		// inflate the shape's stream (zlib, window 15) in one go
		std::vector<unsigned char> data;
		{
			z_stream zs;
			std::memset(&zs, 0, sizeof(zs));
			if (inflateInit2(&zs, 15) != Z_OK)
				fail(PRGPU_EINVAL, path + ": could not initialise zlib");
			zs.next_in	= file.data() + start + 4;
			zs.avail_in = (uInt)(end - start - 4);
			unsigned char buf[1 << 15];
			int rc = Z_OK;
			while (rc == Z_OK) {
				zs.next_out	 = buf;
				zs.avail_out = sizeof(buf);
				rc			 = inflate(&zs, Z_NO_FLUSH);
				if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
					inflateEnd(&zs);
					fail(PRGPU_EINVAL, path + ": zlib data error in shape " + std::to_string(shape));
				}
				data.insert(data.end(), buf, buf + (sizeof(buf) - zs.avail_out));
				if (rc == Z_BUF_ERROR && zs.avail_in == 0)
					break;
			}
			inflateEnd(&zs);
		}
		size_t pos = 0;
		auto take = [&](void* dst, size_t n) {
			if (pos + n > data.size())
				fail(PRGPU_EINVAL, path + ": attempting to read past the end of the stream");
			std::memcpy(dst, data.data() + pos, n);
			pos += n;
		};
		enum { MF_VERTEXNORMALS = 0x0001, MF_TEXCOORDS = 0x0002, MF_VERTEXCOLORS = 0x0008, MF_DOUBLE = 0x2000 };
		uint32_t flags = 0;
		take(&flags, 4);
		if (version >= 4) {
			unsigned char c = 1;
			while (c != 0)
				take(&c, 1); // shape name, ignored
		}
		uint64_t n_vertices = 0, n_tris = 0;
		take(&n_vertices, 8);
		take(&n_tris, 8);
		if (n_vertices == 0 || n_tris == 0 || n_vertices > 0xFFFFFFFFull || n_tris > 0x7FFFFFFFull)
			fail(PRGPU_EINVAL, path + ": no valid mesh in shape " + std::to_string(shape));
		{ // the counts are from the file: check them against the inflated stream before anything is sized by them
			const uint64_t elem = (flags & MF_DOUBLE) ? 8 : 4;
			uint64_t need		= n_vertices * 3 * elem + n_tris * 3 * 4;
			if (flags & MF_VERTEXNORMALS)
				need += n_vertices * 3 * elem;
			if (flags & MF_TEXCOORDS)
				need += n_vertices * 2 * elem;
			if (flags & MF_VERTEXCOLORS)
				need += n_vertices * 3 * elem;
			if (need > data.size() - pos)
				fail(PRGPU_EINVAL, path + ": shape " + std::to_string(shape) + " announces " + std::to_string(n_vertices) + " vertices and " + std::to_string(n_tris)
										+ " triangles but its stream holds " + std::to_string(data.size() - pos) + " bytes");
		}
		auto floats = [&](std::vector<float>& dst, size_t n) {
			dst.resize(n);
			if (flags & MF_DOUBLE) {
				for (size_t i = 0; i < n; ++i) {
					double d;
					take(&d, 8);
					dst[i] = static_cast<float>(d);
				}
			} else {
				take(dst.data(), n * 4);
			}
		};
		Mesh m;
		floats(m.p, n_vertices * 3);
		if (flags & MF_VERTEXNORMALS)
			floats(m.n, n_vertices * 3);
		if (flags & MF_TEXCOORDS)
			floats(m.uv, n_vertices * 2);
		if (flags & MF_VERTEXCOLORS) {
			std::vector<float> ignored;
			floats(ignored, n_vertices * 3);
		}
		for (uint64_t t = 0; t < n_tris; ++t) {
			uint32_t tri[3];
			take(tri, 12);
			for (uint32_t ind : tri)
				if (ind >= n_vertices)
					fail(PRGPU_EINVAL, path + ": triangle index out of range");
			m.faces.push_back({ tri[0], tri[1], tri[2] });
		}
		if (!(flags & MF_VERTEXNORMALS)) { // MeshBase::buildSmoothNormals (src/core/mesh/MeshBase.cpp:15-60): the LAST face of a vertex wins
			m.n.assign(n_vertices * 3, 0.0f);
			for (const auto& face : m.faces) {
				const float *p0 = &m.p[3 * face[0]], *p1 = &m.p[3 * face[1]], *p2 = &m.p[3 * face[2]];
				const float a[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] }, b[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
				float n[3]		 = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
				const float z	 = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
				if (z > 0) { // Triangle::normal = cross.normalized() (Eigen leaves a zero vector alone)
					const float l = std::sqrt(z);
					for (float& c : n)
						c /= l;
				}
				for (uint32_t ind : face)
					for (int c = 0; c < 3; ++c)
						m.n[3 * ind + c] = n[c];
			}
			for (uint64_t i = 0; i < n_vertices; ++i) {
				float* n	  = &m.n[3 * i];
				const float z = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
				if (z > 0) {
					const float l = std::sqrt(z);
					for (int c = 0; c < 3; ++c)
						n[c] /= l;
				}
			}
		}
		const std::string name = get_string(g, "name", "");
		if (name.empty())
			fail(PRGPU_EINVAL, where(g) + ": embedded mesh has no name");
		meshes[name] = std::move(m);
	}

	void add_include(const Group& g, const std::string& dir) // SceneLoader.cpp:848-886
	{
		if (g.anonymous_count() != 1 || g.at(0).type != Value::STRING)
			fail(PRGPU_EINVAL, where(g) + ": include needs one file name");
		if (++include_depth > 16)
			fail(PRGPU_EINVAL, where(g) + ": includes nested too deeply");
		std::string path = g.at(0).s;
		if (!path.empty() && path[0] != '/' && !dir.empty())
			path = dir + "/" + path;
		std::ifstream f(path);
		if (!f)
			fail(PRGPU_EINVAL, where(g) + ": cannot open include file '" + path + "'");
		std::stringstream ss;
		ss << f.rdbuf();
		std::vector<std::shared_ptr<Group>> top;
		std::string err;
		if (!prgpu_host::dl::parse(ss.str(), top, err))
			fail(PRGPU_EINVAL, path + ": " + err);
		const size_t slash = path.find_last_of('/');
		const std::string sub = slash == std::string::npos ? std::string() : path.substr(0, slash);
		for (const auto& t : top)
			dispatch(*t, sub);
		--include_depth;
	}
	void dispatch(const Group& b, const std::string& dir) // SceneLoader.cpp:162-198
	{
		current_dir = dir;
		const std::string& id = b.id;
		if (id == "include")
			add_include(b, dir);
		else if (id == "sampler")
			add_sampler(b);
		else if (id == "filter")
			add_filter(b);
		else if (id == "integrator")
			add_integrator(b);
		else if (id == "spectral_mapper")
			add_mapper(b);
		else if (id == "mesh")
			add_mesh(b);
		else if (id == "material")
			add_material(b);
		else if (id == "emission")
			add_emission(b);
		else if (id == "entity")
			add_entity(b);
		else if (id == "camera")
			add_camera(b);
		else if (id == "output")
			add_output(b);
		else if (id == "embed" || id == "graph")
			add_embed(b, dir);
		else if (id == "light")
			add_light(b);
		else if (id == "texture" || id == "node")
			fail(PRGPU_EUNSUPPORTED, where(b) + ": block is not supported by this backend yet");
		else if (id == "scene")
			fail(PRGPU_EINVAL, where(b) + ": invalid inner scene entry");
		else
			warn(where(b) + ": unknown block ignored"); // the reference's setupEnvironment ignores unknown ids too
	}

	void run(const std::string& source, const std::string& dir)
	{
		std::vector<std::shared_ptr<Group>> top;
		std::string err;
		if (!prgpu_host::dl::parse(source, top, err))
			fail(PRGPU_EINVAL, err);
		if (top.empty() || top.front()->id != "scene")
			fail(PRGPU_EINVAL, "the file does not contain a top-level (scene ...) entry"); // SceneLoader.cpp:76-84
		const Group& scene = *top.front();
		// top-level keys, SceneLoader.cpp:86-140
		if (const Value* v = scene.get("render_width"))
			if (v->type == Value::INT)
				settings.width = (uint32_t)v->i;
		if (const Value* v = scene.get("render_height"))
			if (v->type == Value::INT)
				settings.height = (uint32_t)v->i;
		if (scene.get("crop"))
			fail(PRGPU_EUNSUPPORTED, ":crop is not supported (use tiles)");
		if (const Value* v = scene.get("spectral_domain")) {
			if (v->is_number()) {
				settings.spectral_start = settings.spectral_end = (float)v->number();
				settings.spectral_mono							= 1;
			} else if (v->type == Value::GROUP && v->g->anonymous_count() == 2 && v->g->all_numbers()) {
				settings.spectral_start = (float)std::min(v->g->at(0).number(), v->g->at(1).number());
				settings.spectral_end	= (float)std::max(v->g->at(0).number(), v->g->at(1).number());
				settings.spectral_mono	= settings.spectral_start == settings.spectral_end ? 1 : 0;
			}
		}
		if (const Value* v = scene.get("spectral_hero"))
			if (v->type == Value::BOOL)
				settings.spectral_hero = v->b ? 1 : 0;
		selected_camera = get_string(scene, "camera", "");
		for (const auto& e : scene.entries)
			if (e.key.empty() && e.value.type == Value::GROUP)
				dispatch(*e.value.g, dir);

		// ---- assemble the description ---------------------------------------------------------------------------
		if (opt.width)
			settings.width = opt.width;
		if (opt.height)
			settings.height = opt.height;
		if (opt.aa_samples)
			settings.aa_samples = opt.aa_samples;
		if (opt.seed)
			settings.seed = opt.seed;
		if (cameras.empty())
			fail(PRGPU_EINVAL, "the scene has no camera");
		const std::string cam = selected_camera.empty() ? first_camera : selected_camera;
		if (!cameras.count(cam))
			fail(PRGPU_EINVAL, "the scene's :camera '" + cam + "' does not exist");
		if (out.entities.empty()) {
			// A scene without entities is legal (examples/skylens.prc: a sky seen through a fisheye lens; every camera ray leaves the
			// scene, Scene.cpp:107-118 keeps the default bounding sphere).  The backend's tree wants at least one primitive: one
			// zero-area triangle at the origin, without a material, which no ray can hit.  The scene radius only scales the
			// selection weight of ALL infinite lights alike (LightSampler.cpp:62-71), so it does not matter that it is 0 here.
			prgpu_entity e;
			std::memset(&e, 0, sizeof(e));
			identity(e.transform);
			e.kind		= PRGPU_ENTITY_MESH;
			e.first_tri = 0;
			e.n_tris	= 1;
			e.emission	= PRGPU_INVALID_ID;
			out.positions.assign(9, 0.0f);
			out.indices.assign(3, 0u);
			out.indices[1] = 1;
			out.indices[2] = 2;
			out.tri_material.assign(1, PRGPU_INVALID_ID);
			out.entities.push_back(e);
			warn("the scene has no entities: every camera ray sees the background");
		}
		if (any_normals)
			out.normals.resize(out.positions.size(), 0.0f);
		if (any_uvs)
			out.uvs.resize(out.positions.size() / 3 * 2, 0.0f);
		const uint32_t n_table_values = (uint32_t)out.tables.size();
		if (out.tables.empty())
			out.tables.push_back(0.0f); // keep the pointer valid
		prgpu_scene_desc& d = out.desc;
		std::memset(&d, 0, sizeof(d));
		d.api_version			  = PRGPU_API_VERSION;
		d.n_vertices			  = (uint32_t)(out.positions.size() / 3);
		d.positions				  = out.positions.data();
		d.normals				  = any_normals ? out.normals.data() : nullptr;
		d.uvs					  = any_uvs ? out.uvs.data() : nullptr;
		d.n_triangles			  = (uint32_t)(out.indices.size() / 3);
		d.indices				  = out.indices.data();
		d.tri_material			  = out.tri_material.data();
		d.n_entities			  = (uint32_t)out.entities.size();
		d.entities				  = out.entities.data();
		d.n_materials			  = (uint32_t)out.materials.size();
		d.materials				  = out.materials.data();
		d.n_emissions			  = (uint32_t)out.emissions.size();
		d.emissions				  = out.emissions.data();
		d.n_spectra				  = (uint32_t)out.spectra.size();
		d.spectra				  = out.spectra.data();
		d.n_spectral_table_values = n_table_values;
		d.spectral_tables		  = out.tables.data();
		d.camera				  = cameras[cam];
		d.settings				  = settings;
		d.n_lights				  = (uint32_t)out.lights.size();
		d.lights				  = out.lights.empty() ? nullptr : out.lights.data();
	}
};

int load(const std::string& source, const std::string& dir, const prgpu_prc_options* opt, prgpu_prc** out)
{
	if (!out)
		return PRGPU_EINVAL;
	*out	= nullptr;
	auto* p = new prgpu_prc();
	try {
		Loader l(*p, opt);
		l.run(source, dir);
	} catch (const LoadError& e) {
		g_prc_error = e.msg;
		delete p;
		return e.code;
	} catch (const std::exception& e) {
		g_prc_error = e.what();
		delete p;
		return PRGPU_EINVAL;
	}
	*out = p;
	return PRGPU_OK;
}

} // namespace

extern "C" {

int prgpu_prc_load_string(const char* source, const char* include_dir, const prgpu_prc_options* opt, prgpu_prc** out)
{
	if (!source) {
		g_prc_error = "null source";
		return PRGPU_EINVAL;
	}
	return load(source, include_dir ? include_dir : "", opt, out);
}

int prgpu_prc_load_file(const char* path, const prgpu_prc_options* opt, prgpu_prc** out)
{
	if (!path) {
		g_prc_error = "null path";
		return PRGPU_EINVAL;
	}
	std::ifstream f(path);
	if (!f) {
		g_prc_error = std::string("cannot open '") + path + "'";
		return PRGPU_EINVAL;
	}
	std::stringstream ss;
	ss << f.rdbuf();
	const std::string p(path);
	const size_t slash = p.find_last_of('/');
	return load(ss.str(), slash == std::string::npos ? std::string() : p.substr(0, slash), opt, out);
}

const prgpu_scene_desc* prgpu_prc_desc(const prgpu_prc* p) { return p ? &p->desc : nullptr; }
const char* prgpu_prc_warnings(const prgpu_prc* p) { return p ? p->warnings.c_str() : ""; }
int prgpu_prc_sky_info(const prgpu_prc* p, uint32_t light, prgpu_sky_params* out)
{
	if (!p || !out || light >= p->lights.size() || p->lights[light].kind != PRGPU_LIGHT_SKY || light >= p->sky_params.size())
		return PRGPU_EINVAL;
	*out = p->sky_params[light];
	return PRGPU_OK;
}
const prgpu_output_channel* prgpu_prc_outputs(const prgpu_prc* scene, uint32_t* n_channels)
{
	if (n_channels)
		*n_channels = scene ? (uint32_t)scene->outputs.size() : 0u;
	return scene && !scene->outputs.empty() ? scene->outputs.data() : nullptr;
}
const char* prgpu_prc_output_name(const prgpu_prc* scene, uint32_t file)
{
	return scene && file < scene->output_names.size() ? scene->output_names[file].c_str() : nullptr;
}
const char* prgpu_prc_last_error(void) { return g_prc_error.c_str(); }
void prgpu_prc_free(prgpu_prc* p) { delete p; }

} // extern "C"
