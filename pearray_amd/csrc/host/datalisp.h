// datalisp.h -- reader for PearRay's scene description syntax (".prc", DataLisp s-expressions).
//
// The reference parses scenes with the external DataLisp library (not vendored in the reference tree); this is an
// independent reader of the same surface syntax as used by examples/*.prc and consumed by
// src/loader/SceneLoader.cpp:44-190:
//     (id :key value ... value ...)      a group with named and anonymous entries
//     [v, v, ...]                        an anonymous array group (commas are optional separators)
//     'text' "text"                      strings;  123  -1.5e3  numbers;  true false  booleans
//     ; ...                              comment to end of line
//     (expr a b c)                       an expression is just a group (e.g. (refl 0.7 0.7 0.7))
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace prgpu_host {
namespace dl {

struct Group;

struct Value {
	enum Type { NONE, INT, FLOAT, BOOL, STRING, GROUP } type = NONE;
	int64_t i = 0;
	double f  = 0.0;
	bool b	  = false;
	std::string s;
	std::shared_ptr<Group> g;

	bool is_number() const { return type == INT || type == FLOAT; }
	double number() const { return type == INT ? (double)i : f; }
};

struct Entry {
	std::string key; // empty for anonymous entries
	Value value;
};

struct Group {
	std::string id; // empty for arrays
	bool is_array = false;
	std::vector<Entry> entries;
	int line = 0;

	const Value* get(const std::string& key) const; // last entry wins, like a key/value map
	size_t anonymous_count() const;
	const Value& at(size_t anonymous_index) const; // NONE value when out of range
	bool all_numbers() const;
};

// Parses `source`; returns the top-level groups.  On a syntax error returns false and sets `error` ("line N: ...").
bool parse(const std::string& source, std::vector<std::shared_ptr<Group>>& top, std::string& error);

} // namespace dl
} // namespace prgpu_host
