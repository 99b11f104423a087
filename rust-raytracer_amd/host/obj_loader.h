// Wavefront OBJ / MTL reader with the semantics the reference gets from
// tobj 3.2.5 under LoadOptions{single_index, triangulate, ignore_points,
// ignore_lines} (src/scene.rs:214-224): one model per o/g group (and per
// usemtl change inside a group), polygons fan-triangulated (0, i, i+1), all
// numbers parsed as f32.  The output is already flat -- three positions and
// three normals per triangle -- which is all Triangle::new reads back out of
// the single-index mesh (src/objects/triangle.rs:59-107).
#pragma once
#include <map>
#include <string>
#include <vector>

namespace rrt_host {

struct ObjMaterial {
  std::string name;
  float diffuse[3] = {0, 0, 0};
  float specular[3] = {0, 0, 0};
  float optical_density = 1.0f;
  std::map<std::string, std::string> unknown_param;  // e.g. "Pr", "Tf", "Ke"
};

struct ObjModel {
  std::string name;
  int material_id = -1;            // index into materials, -1 = none
  std::vector<float> positions;    // [n_tris][3][3]
  std::vector<float> normals;      // [n_tris][3][3]; empty when a face had no vn
  size_t n_tris() const { return positions.size() / 9; }
};

struct ObjFile {
  std::vector<ObjModel> models;
  std::vector<ObjMaterial> materials;
};

// Returns false and sets *err when the file cannot be read or is malformed.
bool load_obj(const std::string& path, ObjFile* out, std::string* err);
bool load_mtl(const std::string& path, std::vector<ObjMaterial>* out, std::string* err);

// Area-weighted smooth vertex normals for a model that has none (vertices are
// welded by exact position).  Not something the reference does -- it indexes
// mesh.normals unconditionally (src/objects/triangle.rs:92) and so cannot load
// such a mesh; synthetic stress scenes use this to make cow.obj / lamp.obj usable.
void generate_smooth_normals(ObjModel* model);

}  // namespace rrt_host
