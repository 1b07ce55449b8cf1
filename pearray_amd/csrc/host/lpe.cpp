// lpe.cpp -- light path expressions compiled to a DFA over the 15 (scattering type, scattering event) symbols of a path token.
//
// Replaces src/core/path/LPE_Parser.cpp (grammar), LPE_RegExpr.cpp (Thompson NFA + subset construction) and LPE_Automaton.cpp (the
// per-state transition table) of the reference.  Grammar (LPE_Parser.cpp:66-262):
//     full    := 'C' expr                       the camera vertex, then at least one term
//     expr    := term+
//     term    := (token | '(' expr ')' | '[' term+ ']') op?          [ ... ] is a union; '[^' (negation) is rejected like in the reference
//     token   := D | S | E | L | B | R | T | '.' | '<' type ','? event (','? '"label"')? '>'
//     op      := '*' | '+' | '?' | '{' n '}' | '{' n ',' m '}'
// Token meaning (LPE_RegState.h:39-79): type C camera, E emissive, B background, L emissive or background, R reflection, T refraction,
// '.' reflection or refraction; event D diffuse, S specular, '.' any (including the "none" of camera / light tokens).  D and S alone are
// <.,D> and <.,S>.  A labelled token matches only path tokens with that label; the `direct` integrator never labels its tokens, so a
// labelled token is parsed and matches nothing (as in the reference for this path).
// A path matches when the automaton, fed every token from the camera on, ends in an accepting state (LPE_Automaton.h:17-33).
#include <algorithm>
#include <cctype>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "setup.h"

namespace prgpu_host {
namespace {

struct Node { // regex syntax tree
	enum Kind { TOKEN, CONCAT, UNION, REPEAT } kind = TOKEN;
	uint16_t symbols = 0; // TOKEN: bit s set = symbol s (type * 3 + event) matches
	uint32_t lo = 0, hi = 0; // REPEAT: lo..hi times, hi == 0 = unbounded
	std::vector<std::unique_ptr<Node>> kids;
};

uint16_t symbol_set(char type, char event)
{
	uint16_t m = 0;
	for (int t = 0; t < 5; ++t) { // ScatteringType: Camera, Emissive, Refraction, Reflection, Background (LightPathToken.h:6-13)
		bool mt;
		switch (type) {
		case 'C': mt = t == 0; break;
		case 'E': mt = t == 1; break;
		case 'B': mt = t == 4; break;
		case 'L': mt = t == 1 || t == 4; break;
		case 'R': mt = t == 3; break;
		case 'T': mt = t == 2; break;
		default: mt = t == 2 || t == 3; break; // '.'
		}
		for (int e = 0; e < 3; ++e) { // ScatteringEvent: Diffuse, Specular, None
			const bool me = event == 'D' ? e == 0 : (event == 'S' ? e == 1 : true);
			if (mt && me)
				m |= uint16_t(1u << (t * 3 + e));
		}
	}
	return m;
}

struct Parser {
	const std::string& s;
	size_t pos = 0;
	std::string err;
	int unsupported = 0;
	explicit Parser(const std::string& str) : s(str) {}
	char cur() const { return pos < s.size() ? s[pos] : '\0'; }
	bool eos() const { return pos >= s.size(); }
	bool fail(const std::string& m)
	{
		if (err.empty())
			err = "LPE syntax error at " + std::to_string(pos) + ": " + m;
		return false;
	}
	bool accept(char c)
	{
		if (cur() != c)
			return fail(std::string("expected '") + c + "'");
		++pos;
		return true;
	}
	std::unique_ptr<Node> token_node(char t, char e)
	{
		auto n	   = std::make_unique<Node>();
		n->symbols = symbol_set(t, e);
		return n;
	}
	std::unique_ptr<Node> parse_token()
	{
		const char c = cur();
		if (c == 'D' || c == 'S') {
			++pos;
			return token_node('.', c);
		}
		if (c == 'E' || c == 'L' || c == 'B' || c == 'R' || c == 'T' || c == '.') {
			++pos;
			return token_node(c, '.');
		}
		if (c == '<') {
			++pos;
			const char t = cur();
			if (!(t == 'E' || t == 'L' || t == 'B' || t == 'R' || t == 'T' || t == '.')) {
				fail("expected a scattering type");
				return nullptr;
			}
			++pos;
			if (cur() == ',')
				++pos;
			const char e = cur();
			if (!(e == 'D' || e == 'S' || e == '.')) {
				fail("expected a scattering event");
				return nullptr;
			}
			++pos;
			if (cur() == '"' || cur() == ',') {
				if (cur() == ',')
					++pos;
				if (!accept('"'))
					return nullptr;
				while (!eos() && cur() != '"')
					++pos;
				if (!accept('"'))
					return nullptr;
				if (!accept('>'))
					return nullptr;
				// A labelled token only matches path tokens that carry the same label (LPE_Automaton.cpp:92-110); the `direct` integrator
				// builds every token with label index 0 (LightPathToken.h:38,45; no caller passes one), so on this path it matches nothing.
				std::unique_ptr<Node> dead = token_node(t, e);
				if (dead)
					dead->symbols = 0;
				return dead;
			}
			if (!accept('>'))
				return nullptr;
			return token_node(t, e);
		}
		fail("unknown token");
		return nullptr;
	}
	bool parse_uint(uint32_t& v)
	{
		if (!std::isdigit((unsigned char)cur()))
			return fail("expected a number");
		uint64_t n = 0;
		while (std::isdigit((unsigned char)cur())) {
			n = n * 10 + uint64_t(cur() - '0');
			if (n > 64)
				return fail("repetition counts above 64 are not supported");
			++pos;
		}
		v = (uint32_t)n;
		return true;
	}
	std::unique_ptr<Node> parse_op(std::unique_ptr<Node> n)
	{
		if (!n)
			return nullptr;
		uint32_t lo = 1, hi = 1;
		if (cur() == '*') {
			++pos;
			lo = 0;
			hi = 0;
		} else if (cur() == '+') {
			++pos;
			lo = 1;
			hi = 0;
		} else if (cur() == '?') {
			++pos;
			lo = 0;
			hi = 1;
		} else if (cur() == '{') {
			++pos;
			if (!parse_uint(lo))
				return nullptr;
			hi = lo;
			if (cur() == ',') {
				++pos;
				if (!parse_uint(hi))
					return nullptr;
			}
			if (!accept('}'))
				return nullptr;
			if (hi < lo) {
				fail("maximum less than minimum");
				return nullptr;
			}
			if (lo == 0 && hi == 0) // RegExpr::repeatLast(0, 0) is the star
				hi = 0;
		} else {
			return n;
		}
		auto r	= std::make_unique<Node>();
		r->kind = Node::REPEAT;
		r->lo	= lo;
		r->hi	= hi;
		r->kids.push_back(std::move(n));
		return r;
	}
	std::unique_ptr<Node> parse_term()
	{
		const char c = cur();
		if (c == '(') {
			++pos;
			auto e = parse_expr();
			if (!e || !accept(')'))
				return nullptr;
			return parse_op(std::move(e));
		}
		if (c == '[') {
			++pos;
			if (cur() == '^') {
				fail("negation in union groups is not supported (nor by the reference, LPE_Parser.cpp:148-153)");
				return nullptr;
			}
			auto u	= std::make_unique<Node>();
			u->kind = Node::UNION;
			do {
				auto t = parse_term();
				if (!t)
					return nullptr;
				u->kids.push_back(std::move(t));
			} while (!eos() && cur() != ']');
			if (!accept(']'))
				return nullptr;
			return parse_op(std::move(u));
		}
		if (c == 'D' || c == 'S' || c == 'E' || c == 'L' || c == 'B' || c == 'R' || c == 'T' || c == '.' || c == '<')
			return parse_op(parse_token());
		fail("expected a token or a group");
		return nullptr;
	}
	std::unique_ptr<Node> parse_expr()
	{
		auto cat  = std::make_unique<Node>();
		cat->kind = Node::CONCAT;
		do {
			auto t = parse_term();
			if (!t)
				return nullptr;
			cat->kids.push_back(std::move(t));
		} while (!eos() && cur() != ')');
		return cat;
	}
	std::unique_ptr<Node> parse_full()
	{
		if (!accept('C'))
			return nullptr;
		auto cat  = std::make_unique<Node>();
		cat->kind = Node::CONCAT;
		cat->kids.push_back(token_node('C', '.'));
		auto e = parse_expr();
		if (!e)
			return nullptr;
		if (!eos()) {
			fail("unbalanced ')'");
			return nullptr;
		}
		cat->kids.push_back(std::move(e));
		return cat;
	}
};

// Thompson construction: NFA states with epsilon edges and symbol-set edges
struct Nfa {
	struct Edge {
		int to;
		uint16_t symbols; // 0 = epsilon
	};
	std::vector<std::vector<Edge>> adj;
	int add()
	{
		adj.emplace_back();
		return (int)adj.size() - 1;
	}
};
struct Frag {
	int in, out;
};
Frag build(const Node& n, Nfa& a)
{
	switch (n.kind) {
	case Node::TOKEN: {
		const int i = a.add(), o = a.add();
		if (n.symbols != 0) // an empty symbol set (labelled token) is a dead end, not an epsilon edge
			a.adj[i].push_back({ o, n.symbols });
		return { i, o };
	}
	case Node::CONCAT: {
		Frag f = build(*n.kids[0], a);
		for (size_t k = 1; k < n.kids.size(); ++k) {
			const Frag g = build(*n.kids[k], a);
			a.adj[f.out].push_back({ g.in, 0 });
			f.out = g.out;
		}
		return f;
	}
	case Node::UNION: {
		const int i = a.add(), o = a.add();
		for (const auto& k : n.kids) {
			const Frag g = build(*k, a);
			a.adj[i].push_back({ g.in, 0 });
			a.adj[g.out].push_back({ o, 0 });
		}
		return { i, o };
	}
	default: { // REPEAT lo..hi (hi == 0: unbounded)
		const int i = a.add();
		int cur		= i;
		for (uint32_t k = 0; k < n.lo; ++k) {
			const Frag g = build(*n.kids[0], a);
			a.adj[cur].push_back({ g.in, 0 });
			cur = g.out;
		}
		const int o = a.add();
		if (n.hi == 0) { // then any number more
			const Frag g = build(*n.kids[0], a);
			a.adj[cur].push_back({ g.in, 0 });
			a.adj[g.out].push_back({ g.in, 0 });
			a.adj[g.out].push_back({ o, 0 });
			a.adj[cur].push_back({ o, 0 });
		} else {
			a.adj[cur].push_back({ o, 0 });
			for (uint32_t k = n.lo; k < n.hi; ++k) { // up to hi - lo optional ones
				const Frag g = build(*n.kids[0], a);
				a.adj[cur].push_back({ g.in, 0 });
				a.adj[g.out].push_back({ o, 0 });
				cur = g.out;
			}
		}
		return { i, o };
	}
	}
}
void closure(const Nfa& a, std::set<int>& st)
{
	std::vector<int> work(st.begin(), st.end());
	while (!work.empty()) {
		const int s = work.back();
		work.pop_back();
		for (const auto& e : a.adj[s])
			if (e.symbols == 0 && st.insert(e.to).second)
				work.push_back(e.to);
	}
}

} // namespace

// DFA of `expr`: next[state * 15 + symbol] = following state or 0xFF (the path can no longer match), accepting[state]; state 0 is the
// start.  At most PRGPU_LPE_MAX_STATES states.  Returns PRGPU_OK, PRGPU_EINVAL (syntax) or PRGPU_EUNSUPPORTED (too many states).
int compile_lpe(const std::string& expr, std::vector<uint8_t>& next, std::vector<uint8_t>& accepting, std::string& err)
{
	Parser p(expr);
	const auto tree = p.parse_full();
	if (!tree) {
		err = p.err.empty() ? "invalid light path expression" : p.err;
		return p.unsupported ? PRGPU_EUNSUPPORTED : PRGPU_EINVAL;
	}
	Nfa nfa;
	const Frag f = build(*tree, nfa);
	std::map<std::set<int>, int> ids;
	std::vector<std::set<int>> states;
	std::set<int> start{ f.in };
	closure(nfa, start);
	ids[start] = 0;
	states.push_back(start);
	next.clear();
	accepting.clear();
	for (size_t si = 0; si < states.size(); ++si) {
		next.resize((si + 1) * 15, 0xFF);
		accepting.push_back(states[si].count(f.out) ? 1 : 0);
		for (int sym = 0; sym < 15; ++sym) {
			std::set<int> to;
			for (int q : states[si])
				for (const auto& e : nfa.adj[q])
					if (e.symbols & (1u << sym))
						to.insert(e.to);
			if (to.empty())
				continue;
			closure(nfa, to);
			auto it = ids.find(to);
			if (it == ids.end()) {
				if (states.size() >= PRGPU_LPE_MAX_STATES) {
					err = "light path expression needs more than " + std::to_string(PRGPU_LPE_MAX_STATES) + " automaton states";
					return PRGPU_EUNSUPPORTED;
				}
				it = ids.emplace(to, (int)states.size()).first;
				states.push_back(to);
			}
			next[si * 15 + sym] = (uint8_t)it->second;
		}
	}
	return PRGPU_OK;
}

} // namespace prgpu_host
