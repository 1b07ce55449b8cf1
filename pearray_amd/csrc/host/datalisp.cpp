// datalisp.cpp -- recursive-descent reader for the .prc surface syntax (see datalisp.h).
#include "datalisp.h"

#include <cctype>
#include <cstdlib>

namespace prgpu_host {
namespace dl {

const Value* Group::get(const std::string& key) const
{
	const Value* found = nullptr;
	for (const Entry& e : entries)
		if (e.key == key)
			found = &e.value;
	return found;
}
size_t Group::anonymous_count() const
{
	size_t n = 0;
	for (const Entry& e : entries)
		n += e.key.empty();
	return n;
}
const Value& Group::at(size_t index) const
{
	static const Value none;
	for (const Entry& e : entries)
		if (e.key.empty()) {
			if (index == 0)
				return e.value;
			--index;
		}
	return none;
}
bool Group::all_numbers() const
{
	for (const Entry& e : entries)
		if (e.key.empty() && !e.value.is_number())
			return false;
	return true;
}

namespace {

struct Reader {
	const std::string& src;
	size_t pos = 0;
	int line   = 1;
	int depth  = 0;
	static constexpr int MAX_DEPTH = 256; // nesting of ( and [ groups
	std::string error;

	explicit Reader(const std::string& s) : src(s) {}

	bool fail(const std::string& msg)
	{
		if (error.empty())
			error = "line " + std::to_string(line) + ": " + msg;
		return false;
	}
	void skip()
	{
		while (pos < src.size()) {
			const char c = src[pos];
			if (c == '\n') {
				++line;
				++pos;
			} else if (std::isspace((unsigned char)c) || c == ',') {
				++pos;
			} else if (c == ';') {
				while (pos < src.size() && src[pos] != '\n')
					++pos;
			} else {
				break;
			}
		}
	}
	static bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == '.' || c == '+'; }

	bool read_string(std::string& out)
	{
		const char quote = src[pos++];
		out.clear();
		while (pos < src.size() && src[pos] != quote) {
			char c = src[pos++];
			if (c == '\n')
				++line;
			if (c == '\\' && pos < src.size()) {
				const char n = src[pos++];
				c = n == 'n' ? '\n' : n == 't' ? '\t' : n;
			}
			out.push_back(c);
		}
		if (pos >= src.size())
			return fail("unterminated string");
		++pos;
		return true;
	}
	bool read_value(Value& v)
	{
		skip();
		if (pos >= src.size())
			return fail("unexpected end of input");
		const char c = src[pos];
		if (c == '(' || c == '[') {
			v.type = Value::GROUP;
			v.g	   = std::make_shared<Group>();
			return read_group(*v.g);
		}
		if (c == '\'' || c == '"') {
			v.type = Value::STRING;
			return read_string(v.s);
		}
		if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
			const char* begin = src.c_str() + pos;
			char* end_i		  = nullptr;
			char* end_f		  = nullptr;
			const long long iv = std::strtoll(begin, &end_i, 10);
			const double fv	   = std::strtod(begin, &end_f);
			if (end_f == begin)
				return fail(std::string("invalid number near '") + c + "'");
			if (end_i == end_f) {
				v.type = Value::INT;
				v.i	   = iv;
			} else {
				v.type = Value::FLOAT;
				v.f	   = fv;
			}
			pos += (size_t)(end_f - begin);
			return true;
		}
		if (std::isalpha((unsigned char)c) || c == '_') {
			std::string word;
			while (pos < src.size() && ident_char(src[pos]))
				word.push_back(src[pos++]);
			if (word == "true" || word == "false") {
				v.type = Value::BOOL;
				v.b	   = word == "true";
			} else { // a bare word is treated as a string (DataLisp accepts unquoted identifiers as values)
				v.type = Value::STRING;
				v.s	   = word;
			}
			return true;
		}
		return fail(std::string("unexpected character '") + c + "'");
	}
	bool read_group(Group& g)
	{
		// a hostile or broken file must end in a parse error, not in a host stack overflow
		struct DepthGuard {
			int& d;
			explicit DepthGuard(int& d_) : d(d_) { ++d; }
			~DepthGuard() { --d; }
		} guard(depth);
		if (depth > MAX_DEPTH)
			return fail("nesting too deep");
		const char open	 = src[pos++];
		const char close = open == '(' ? ')' : ']';
		g.is_array		 = open == '[';
		g.line			 = line;
		if (!g.is_array) {
			skip();
			while (pos < src.size() && ident_char(src[pos]))
				g.id.push_back(src[pos++]);
			if (g.id.empty())
				return fail("group without an identifier");
		}
		for (;;) {
			skip();
			if (pos >= src.size())
				return fail(std::string("missing '") + close + "' for group opened at line " + std::to_string(g.line));
			if (src[pos] == close) {
				++pos;
				return true;
			}
			if (src[pos] == ')' || src[pos] == ']')
				return fail("mismatched bracket");
			Entry e;
			if (src[pos] == ':') {
				if (g.is_array)
					return fail("keys are not allowed inside arrays");
				++pos;
				while (pos < src.size() && ident_char(src[pos]))
					e.key.push_back(src[pos++]);
				if (e.key.empty())
					return fail("empty key");
			}
			if (!read_value(e.value))
				return false;
			g.entries.push_back(std::move(e));
		}
	}
};

} // namespace

bool parse(const std::string& source, std::vector<std::shared_ptr<Group>>& top, std::string& error)
{
	Reader r(source);
	top.clear();
	for (;;) {
		r.skip();
		if (r.pos >= source.size())
			return true;
		if (source[r.pos] != '(') {
			r.fail(std::string("expected '(' at top level, found '") + source[r.pos] + "'");
			error = r.error;
			return false;
		}
		auto g = std::make_shared<Group>();
		if (!r.read_group(*g)) {
			error = r.error;
			return false;
		}
		top.push_back(g);
	}
}

} // namespace dl
} // namespace prgpu_host
