#include "yaml_lite.h"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rrt_host {

namespace {

const Yaml kBad;

struct Line {
  int indent;
  std::string text;
  int number;
};

std::string rstrip(const std::string& s) {
  size_t e = s.size();
  while (e > 0 && (s[e - 1] == ' ' || s[e - 1] == '\t' || s[e - 1] == '\r')) --e;
  return s.substr(0, e);
}

std::string strip(const std::string& s) {
  size_t b = 0;
  while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) ++b;
  return rstrip(s.substr(b));
}

// Removes a trailing comment: '#' at line start or preceded by whitespace, outside quotes.
std::string strip_comment(const std::string& s) {
  char quote = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    char c = s[i];
    if (quote) {
      if (c == quote) quote = 0;
    } else if (c == '"' || c == '\'') {
      quote = c;
    } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
      return s.substr(0, i);
    }
  }
  return s;
}

bool parse_int(const std::string& v, long long* out) {
  if (v.empty()) return false;
  size_t i = 0;
  if (v[0] == '+' || v[0] == '-') i = 1;
  if (i >= v.size()) return false;
  for (size_t k = i; k < v.size(); ++k)
    if (v[k] < '0' || v[k] > '9') return false;
  errno = 0;
  char* end = nullptr;
  long long x = std::strtoll(v.c_str(), &end, 10);
  if (errno || *end) return false;
  *out = x;
  return true;
}

bool parse_real(const std::string& v, double* out) {
  if (v.empty()) return false;
  if (v == ".inf" || v == ".Inf" || v == ".INF" || v == "+.inf") { *out = 1.0 / 0.0; return true; }
  if (v == "-.inf" || v == "-.Inf" || v == "-.INF") { *out = -1.0 / 0.0; return true; }
  // digits, sign, '.', exponent only: "y", "none" etc. must stay strings
  bool digit = false;
  for (char c : v) {
    if (c >= '0' && c <= '9') digit = true;
    else if (c != '+' && c != '-' && c != '.' && c != 'e' && c != 'E') return false;
  }
  if (!digit) return false;
  char* end = nullptr;
  errno = 0;
  double x = std::strtod(v.c_str(), &end);
  if (*end) return false;
  *out = x;
  return true;
}

Yaml make_scalar(const std::string& raw) {
  Yaml y;
  std::string v = strip(raw);
  if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) {
    y.kind = Yaml::String;
    y.text = v.substr(1, v.size() - 2);
    return y;
  }
  y.text = v;
  long long i;
  double d;
  if (v.empty() || v == "~" || v == "null") y.kind = Yaml::Null;
  else if (v == "true" || v == "false") y.kind = Yaml::Boolean;
  else if (parse_int(v, &i)) y.kind = Yaml::Integer;
  else if (parse_real(v, &d)) y.kind = Yaml::Real;
  else y.kind = Yaml::String;
  return y;
}

bool parse_flow(const std::string& v, Yaml* out, std::string* err) {
  std::string s = strip(v);
  if (!s.empty() && s.front() == '[') {
    if (s.back() != ']') { *err = "unterminated flow sequence: " + s; return false; }
    out->kind = Yaml::Seq;
    std::string body = s.substr(1, s.size() - 2);
    std::string item;
    int depth = 0;
    for (size_t i = 0; i <= body.size(); ++i) {
      char c = i < body.size() ? body[i] : ',';
      if (c == '[') ++depth;
      if (c == ']') --depth;
      if (c == ',' && depth == 0) {
        if (!strip(item).empty()) {
          Yaml child;
          if (!parse_flow(item, &child, err)) return false;
          out->seq.push_back(child);
        }
        item.clear();
      } else {
        item.push_back(c);
      }
    }
    return true;
  }
  *out = make_scalar(s);
  return true;
}

// Position of the ':' that ends a map key, or npos.
size_t find_key_colon(const std::string& s) {
  char quote = 0;
  int depth = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    char c = s[i];
    if (quote) { if (c == quote) quote = 0; continue; }
    if (c == '"' || c == '\'') { quote = c; continue; }
    if (c == '[' || c == '{') ++depth;
    if (c == ']' || c == '}') --depth;
    if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
  }
  return std::string::npos;
}

struct Parser {
  std::vector<Line> lines;
  size_t pos = 0;
  std::string err;

  bool fail(const Line& l, const std::string& msg) {
    err = "yaml line " + std::to_string(l.number) + ": " + msg;
    return false;
  }

  bool parse_block(int indent, Yaml* out) {
    if (pos >= lines.size()) { out->kind = Yaml::Null; return true; }
    const std::string& first = lines[pos].text;
    if (first == "-" || first.rfind("- ", 0) == 0) return parse_seq(indent, out);
    return parse_map(indent, out);
  }

  bool parse_seq(int indent, Yaml* out) {
    out->kind = Yaml::Seq;
    while (pos < lines.size() && lines[pos].indent == indent &&
           (lines[pos].text == "-" || lines[pos].text.rfind("- ", 0) == 0)) {
      Line& l = lines[pos];
      std::string rest = l.text.size() > 1 ? l.text.substr(2) : std::string();
      size_t lead = 0;
      while (lead < rest.size() && rest[lead] == ' ') ++lead;
      rest = rest.substr(lead);
      Yaml item;
      if (rest.empty()) {
        ++pos;
        if (pos < lines.size() && lines[pos].indent > indent) {
          if (!parse_block(lines[pos].indent, &item)) return false;
        } else {
          item.kind = Yaml::Null;
        }
      } else if (find_key_colon(rest) != std::string::npos && rest.front() != '[') {
        // "- key: value": a map whose keys sit at the column of `key`
        l.indent = indent + 2 + (int)lead;
        l.text = rest;
        if (!parse_map(l.indent, &item)) return false;
      } else {
        ++pos;
        if (!parse_flow(rest, &item, &err)) return false;
      }
      out->seq.push_back(item);
    }
    return true;
  }

  bool parse_map(int indent, Yaml* out) {
    out->kind = Yaml::Map;
    while (pos < lines.size() && lines[pos].indent == indent) {
      const Line l = lines[pos];
      if (l.text == "-" || l.text.rfind("- ", 0) == 0) break;
      size_t colon = find_key_colon(l.text);
      if (colon == std::string::npos) return fail(l, "expected 'key: value'");
      std::string key = strip(l.text.substr(0, colon));
      if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
      std::string rest = strip(l.text.substr(colon + 1));
      ++pos;
      Yaml value;
      if (rest.empty()) {
        if (pos < lines.size() && lines[pos].indent > indent) {
          if (!parse_block(lines[pos].indent, &value)) return false;
        } else if (pos < lines.size() && lines[pos].indent == indent &&
                   (lines[pos].text == "-" || lines[pos].text.rfind("- ", 0) == 0)) {
          if (!parse_seq(indent, &value)) return false;
        } else {
          value.kind = Yaml::Null;
        }
      } else {
        if (!parse_flow(rest, &value, &err)) return false;
      }
      out->map.emplace_back(key, value);
    }
    if (pos < lines.size() && lines[pos].indent > indent) return fail(lines[pos], "unexpected indentation");
    return true;
  }
};

}  // namespace

const Yaml& Yaml::operator[](const std::string& key) const {
  if (kind != Map) return kBad;
  for (const auto& kv : map)
    if (kv.first == key) return kv.second;
  return kBad;
}

const Yaml& Yaml::operator[](size_t i) const {
  if (kind != Seq || i >= seq.size()) return kBad;
  return seq[i];
}

bool Yaml::as_f64(double* out) const {
  if (kind != Real) return false;
  return parse_real(text, out);
}

bool Yaml::as_i64(long long* out) const {
  if (kind != Integer) return false;
  return parse_int(text, out);
}

bool Yaml::as_str(std::string* out) const {
  if (kind != String) return false;
  *out = text;
  return true;
}

bool Yaml::as_bool(bool* out) const {
  if (kind != Boolean) return false;
  *out = text == "true";
  return true;
}

bool Yaml::parse(const std::string& src, Yaml* out, std::string* err) {
  Parser p;
  std::istringstream in(src);
  std::string raw;
  int number = 0;
  while (std::getline(in, raw)) {
    ++number;
    std::string s = rstrip(strip_comment(raw));
    if (strip(s).empty()) continue;
    if (s == "---" && p.lines.empty()) continue;
    int indent = 0;
    while ((size_t)indent < s.size() && s[indent] == ' ') ++indent;
    if ((size_t)indent < s.size() && s[indent] == '\t') {
      *err = "yaml line " + std::to_string(number) + ": tab indentation";
      return false;
    }
    p.lines.push_back({indent, s.substr(indent), number});
  }
  *out = Yaml();
  if (p.lines.empty()) { out->kind = Null; return true; }
  if (!p.parse_block(p.lines[0].indent, out)) { *err = p.err; return false; }
  if (p.pos != p.lines.size()) {
    *err = "yaml line " + std::to_string(p.lines[p.pos].number) + ": unexpected dedent";
    return false;
  }
  return true;
}

bool Yaml::parse_file(const std::string& path, Yaml* out, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { *err = "Unable to open " + path; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  return parse(ss.str(), out, err);
}

}  // namespace rrt_host
