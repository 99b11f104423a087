#include "obj_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace rrt_host {

namespace {

bool read_file(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

inline void skip_ws(const char*& p) {
  while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
}

// One whitespace-delimited token; empty at end of line.
inline std::string next_token(const char*& p) {
  skip_ws(p);
  const char* b = p;
  while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
  return std::string(b, p);
}

inline std::string rest_of_line(const char* p) {
  skip_ws(p);
  const char* e = p;
  while (*e && *e != '\n') ++e;
  while (e > p && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
  return std::string(p, e);
}

bool parse_floats(const char*& p, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    skip_ws(p);
    char* end = nullptr;
    float v = std::strtof(p, &end);
    if (end == p) return false;
    out[i] = v;
    p = end;
  }
  return true;
}

struct Corner {
  long v = 0, vt = 0, vn = 0;  // 1-based after resolution, 0 = absent
};

bool parse_corner(const std::string& tok, size_t nv, size_t nvt, size_t nvn, Corner* c) {
  long idx[3] = {0, 0, 0};
  bool present[3] = {false, false, false};
  size_t field = 0;
  const char* p = tok.c_str();
  while (*p && field < 3) {
    if (*p == '/') { ++field; ++p; continue; }
    char* end = nullptr;
    long x = std::strtol(p, &end, 10);
    if (end == p) return false;
    idx[field] = x;
    present[field] = true;
    p = end;
  }
  if (!present[0]) return false;
  const size_t counts[3] = {nv, nvt, nvn};
  for (int k = 0; k < 3; ++k) {
    if (!present[k]) continue;
    if (idx[k] < 0) idx[k] = (long)counts[k] + idx[k] + 1;  // relative index
    if (idx[k] <= 0 || (size_t)idx[k] > counts[k]) return false;
  }
  c->v = idx[0];
  c->vt = idx[1];
  c->vn = idx[2];
  return true;
}

}  // namespace

bool load_mtl(const std::string& path, std::vector<ObjMaterial>* out, std::string* err) {
  std::string src;
  if (!read_file(path, &src)) { *err = "cannot open " + path; return false; }
  const char* p = src.c_str();
  ObjMaterial* cur = nullptr;
  while (*p) {
    const char* line = p;
    const char* eol = std::strchr(p, '\n');
    p = eol ? eol + 1 : p + std::strlen(p);
    const char* q = line;
    std::string key = next_token(q);
    if (key.empty() || key[0] == '#') continue;
    if (key == "newmtl") {
      out->emplace_back();
      cur = &out->back();
      cur->name = rest_of_line(q);
      continue;
    }
    if (!cur) continue;
    if (key == "Kd") { if (!parse_floats(q, cur->diffuse, 3)) { *err = path + ": bad Kd"; return false; } }
    else if (key == "Ks") { if (!parse_floats(q, cur->specular, 3)) { *err = path + ": bad Ks"; return false; } }
    else if (key == "Ni") { if (!parse_floats(q, &cur->optical_density, 1)) { *err = path + ": bad Ni"; return false; } }
    else if (key == "Ka" || key == "Ns" || key == "d" || key == "illum" || key.rfind("map_", 0) == 0 || key == "bump") {
      // parsed by tobj into fields the reference never reads
    } else {
      cur->unknown_param[key] = rest_of_line(q);
    }
  }
  return true;
}

bool load_obj(const std::string& path, ObjFile* out, std::string* err) {
  std::string src;
  if (!read_file(path, &src)) { *err = "Failed to load file " + path; return false; }
  std::string dir;
  size_t slash = path.find_last_of('/');
  if (slash != std::string::npos) dir = path.substr(0, slash + 1);

  std::vector<float> v, vn;
  size_t n_vt = 0;
  ObjModel cur;
  cur.name = "unnamed_object";
  std::unordered_map<std::string, int> mat_index;
  bool any_missing_normal = false;

  auto flush = [&]() {
    if (!cur.positions.empty()) {
      if (any_missing_normal) cur.normals.clear();
      out->models.push_back(cur);
    }
    cur.positions.clear();
    cur.normals.clear();
    any_missing_normal = false;
  };

  const char* p = src.c_str();
  std::vector<Corner> face;
  size_t line_no = 0;
  while (*p) {
    ++line_no;
    const char* line = p;
    const char* eol = std::strchr(p, '\n');
    p = eol ? eol + 1 : p + std::strlen(p);
    const char* q = line;
    skip_ws(q);
    if (*q == '#' || *q == '\n' || *q == 0) continue;
    if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
      q += 1;
      float xyz[3];
      if (!parse_floats(q, xyz, 3)) { *err = path + ":" + std::to_string(line_no) + ": bad vertex"; return false; }
      v.insert(v.end(), xyz, xyz + 3);
    } else if (q[0] == 'v' && q[1] == 'n') {
      q += 2;
      float xyz[3];
      if (!parse_floats(q, xyz, 3)) { *err = path + ":" + std::to_string(line_no) + ": bad normal"; return false; }
      vn.insert(vn.end(), xyz, xyz + 3);
    } else if (q[0] == 'v' && q[1] == 't') {
      ++n_vt;
    } else if (q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
      q += 1;
      face.clear();
      for (;;) {
        std::string tok = next_token(q);
        if (tok.empty()) break;
        Corner c;
        if (!parse_corner(tok, v.size() / 3, n_vt, vn.size() / 3, &c)) {
          *err = path + ":" + std::to_string(line_no) + ": bad face index '" + tok + "'";
          return false;
        }
        face.push_back(c);
      }
      if (face.size() < 3) continue;  // points / lines are ignored
      for (size_t i = 1; i + 1 < face.size(); ++i) {
        const Corner tri[3] = {face[0], face[i], face[i + 1]};
        for (const Corner& c : tri) {
          const float* pv = &v[3 * (c.v - 1)];
          cur.positions.insert(cur.positions.end(), pv, pv + 3);
          if (c.vn > 0) {
            const float* pn = &vn[3 * (c.vn - 1)];
            cur.normals.insert(cur.normals.end(), pn, pn + 3);
          } else {
            any_missing_normal = true;
            cur.normals.insert(cur.normals.end(), 3, 0.0f);
          }
        }
      }
    } else if ((q[0] == 'o' || q[0] == 'g') && (q[1] == ' ' || q[1] == '\t' || q[1] == '\r' || q[1] == '\n' || q[1] == 0)) {
      flush();
      cur.name = rest_of_line(q + 1);
      if (cur.name.empty()) cur.name = "unnamed_object";
    } else {
      std::string key = next_token(q);
      if (key == "mtllib") {
        std::string file = rest_of_line(q);
        std::vector<ObjMaterial> mats;
        std::string merr;
        if (!load_mtl(dir + file, &mats, &merr)) { *err = merr; return false; }
        for (auto& m : mats) {
          mat_index[m.name] = (int)out->materials.size();
          out->materials.push_back(m);
        }
      } else if (key == "usemtl") {
        std::string name = rest_of_line(q);
        auto it = mat_index.find(name);
        int id = it == mat_index.end() ? -1 : it->second;
        if (id != cur.material_id) {
          std::string keep = cur.name;
          flush();
          cur.name = keep;
        }
        cur.material_id = id;
      }
      // s, l, p, vp, ... ignored
    }
  }
  flush();
  return true;
}

void generate_smooth_normals(ObjModel* model) {
  struct Key {
    float x, y, z;
    bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
  };
  struct Hash {
    size_t operator()(const Key& k) const {
      uint32_t a, b, c;
      std::memcpy(&a, &k.x, 4);
      std::memcpy(&b, &k.y, 4);
      std::memcpy(&c, &k.z, 4);
      return (size_t)a * 73856093u ^ (size_t)b * 19349663u ^ (size_t)c * 83492791u;
    }
  };
  const size_t nt = model->n_tris();
  std::unordered_map<Key, size_t, Hash> index;
  std::vector<double> acc;
  std::vector<size_t> corner(nt * 3);
  for (size_t t = 0; t < nt; ++t) {
    const float* P = &model->positions[9 * t];
    double e1[3], e2[3], n[3];
    for (int k = 0; k < 3; ++k) { e1[k] = (double)P[3 + k] - P[k]; e2[k] = (double)P[6 + k] - P[k]; }
    n[0] = e1[1] * e2[2] - e1[2] * e2[1];
    n[1] = e1[2] * e2[0] - e1[0] * e2[2];
    n[2] = e1[0] * e2[1] - e1[1] * e2[0];  // length = 2 * area: area weighting
    for (int c = 0; c < 3; ++c) {
      Key key{P[3 * c], P[3 * c + 1], P[3 * c + 2]};
      auto it = index.find(key);
      size_t id;
      if (it == index.end()) {
        id = acc.size() / 3;
        index.emplace(key, id);
        acc.insert(acc.end(), 3, 0.0);
      } else {
        id = it->second;
      }
      corner[3 * t + c] = id;
      for (int k = 0; k < 3; ++k) acc[3 * id + k] += n[k];
    }
  }
  model->normals.assign(nt * 9, 0.0f);
  for (size_t i = 0; i < nt * 3; ++i) {
    const double* a = &acc[3 * corner[i]];
    double len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (len > 0) {
      for (int k = 0; k < 3; ++k) model->normals[3 * i + k] = (float)(a[k] / len);
    } else {
      model->normals[3 * i + 1] = 1.0f;
    }
  }
}

}  // namespace rrt_host
