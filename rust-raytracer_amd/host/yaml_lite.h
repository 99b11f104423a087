// Minimal YAML reader for scene.yaml / render_settings.yaml.
//
// The reference reads both files with yaml-rust 0.4.5 (src/scene.rs:57-62,
// src/main.rs:245-250) and relies on that crate's typed accessors: as_f64()
// is Some only for a scalar written as a real ("10.0", not "10"), as_i64()
// only for an integer, as_str() only for a plain string, and indexing a
// missing key or a null yields a "bad value" for which every accessor is None
// (src/main.rs:256,263-275,296).  This reader keeps exactly those
// distinctions for the subset of YAML the shipped files use: block maps, block
// sequences of maps, flow sequences of scalars, comments.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace rrt_host {

struct Yaml {
  enum Kind { Bad, Null, String, Integer, Real, Boolean, Map, Seq };
  Kind kind = Bad;
  std::string text;                                // scalar text
  std::vector<std::pair<std::string, Yaml>> map;   // insertion order
  std::vector<Yaml> seq;

  // yaml["key"] / yaml[i]; a miss returns the shared Bad node.
  const Yaml& operator[](const std::string& key) const;
  const Yaml& operator[](size_t i) const;

  bool is_badvalue() const { return kind == Bad; }
  bool as_f64(double* out) const;   // Real only
  bool as_i64(long long* out) const;  // Integer only
  bool as_str(std::string* out) const;  // String only
  bool as_bool(bool* out) const;
  size_t size() const { return kind == Seq ? seq.size() : (kind == Map ? map.size() : 0); }

  // Parses one document.  On a syntax error returns false and sets *err.
  static bool parse(const std::string& src, Yaml* out, std::string* err);
  static bool parse_file(const std::string& path, Yaml* out, std::string* err);
};

}  // namespace rrt_host
