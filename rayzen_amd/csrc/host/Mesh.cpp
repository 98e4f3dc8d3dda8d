// Mesh.cpp -- OBJ reader with the exact quirks of RayZen/src/Mesh.cpp:6-50:
// only "v " and "f " lines; a face token is cut at its first '/'; indices are
// 1-based and never negative; polygons are fan-triangulated around their
// first vertex; everything else (vn, vt, o, g, s, usemtl ...) is ignored.
#include "RayZenScene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rayzen {

bool Mesh::loadFromOBJ(const std::string& filename, int materialIndex) {
    FILE* f = std::fopen(filename.c_str(), "r");
    if (!f) {
        std::fprintf(stderr, "[ERROR] Failed to open OBJ file: %s\n", filename.c_str());
        return false;
    }
    std::vector<vec3> vertices;
    std::vector<unsigned> face;
    char* line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, f) >= 0) {
        if (line[0] == 'v' && line[1] == ' ') {
            vec3 v;
            char* p = line + 2;
            v.x = std::strtof(p, &p);
            v.y = std::strtof(p, &p);
            v.z = std::strtof(p, &p);
            vertices.push_back(v);
        } else if (line[0] == 'f' && line[1] == ' ') {
            face.clear();
            char* save = nullptr;
            for (char* tok = strtok_r(line + 2, " \t\r\n", &save); tok; tok = strtok_r(nullptr, " \t\r\n", &save)) {
                if (char* slash = std::strchr(tok, '/')) *slash = 0;
                face.push_back((unsigned)std::atoi(tok));
            }
            if (face.size() >= 3) {
                for (size_t i = 1; i + 1 < face.size(); ++i) {
                    unsigned a = face[0] - 1, b = face[i] - 1, c = face[i + 1] - 1;
                    if (a >= vertices.size() || b >= vertices.size() || c >= vertices.size()) continue;  // reference: UB
                    Triangle tri;
                    std::memset(static_cast<void*>(&tri), 0, sizeof tri);   // tail padding travels to the GPU: keep it defined
                    tri.v0 = vertices[a];
                    tri.v1 = vertices[b];
                    tri.v2 = vertices[c];
                    tri.materialIndex = materialIndex;
                    triangles.push_back(tri);
                }
            }
        }
    }
    std::free(line);
    std::fclose(f);
    return true;
}

}  // namespace rayzen
