// SceneCache.cpp -- see SceneCache.h (RayZen/src/main.cpp:94-133, 914-939, 1037-1043).
#include "SceneCache.h"

#include <algorithm>

namespace rayzen {

namespace {
std::string join(const std::string& dir, const char* name) {
    if (dir.empty()) return name;
    return dir.back() == '/' ? dir + name : dir + "/" + name;
}
int subtreeDepth(const std::vector<BVHNode>& nodes, int root) {
    if (root < 0 || root >= (int)nodes.size()) return 0;
    int best = 1;
    std::vector<std::pair<int, int>> st{{root, 1}};
    size_t guard = 0;
    while (!st.empty() && guard++ <= nodes.size() * 2) {
        auto [n, d] = st.back();
        st.pop_back();
        best = std::max(best, d);
        if (nodes[n].count < 0 && nodes[n].leftFirst >= 0 && nodes[n].leftFirst + 1 < (int)nodes.size()) {
            st.push_back({nodes[n].leftFirst, d + 1});
            st.push_back({nodes[n].leftFirst + 1, d + 1});
        }
    }
    return best;
}
}  // namespace

bool saveSceneCache(const std::string& dir, const SceneBuffers& b) {
    return saveVectorToFile(join(dir, "ssbo_v2_triangles.bin"), b.allTriangles) &&
           saveVectorToFile(join(dir, "ssbo_v2_blasnodes.bin"), b.allBLASNodes) &&
           saveVectorToFile(join(dir, "ssbo_v2_blastris.bin"), b.allBLASTriIndices) &&
           saveVectorToFile(join(dir, "ssbo_v2_instances.bin"), b.meshInstances) &&
           saveVectorToFile(join(dir, "ssbo_v2_tlasnodes.bin"), b.tlasNodes) &&
           saveVectorToFile(join(dir, "ssbo_v2_tlastris.bin"), b.tlasTriIndices);
}

bool loadSceneCache(const std::string& dir, SceneBuffers& b) {
    SceneBuffers t;
    if (!(loadVectorFromFile(join(dir, "ssbo_v2_triangles.bin"), t.allTriangles) &&
          loadVectorFromFile(join(dir, "ssbo_v2_blasnodes.bin"), t.allBLASNodes) &&
          loadVectorFromFile(join(dir, "ssbo_v2_blastris.bin"), t.allBLASTriIndices) &&
          loadVectorFromFile(join(dir, "ssbo_v2_instances.bin"), t.meshInstances) &&
          loadVectorFromFile(join(dir, "ssbo_v2_tlasnodes.bin"), t.tlasNodes) &&
          loadVectorFromFile(join(dir, "ssbo_v2_tlastris.bin"), t.tlasTriIndices)))
        return false;
    t.blasRoots.clear();
    t.maxBLASDepth = 1;
    for (const BVHInstance& inst : t.meshInstances) {
        if (inst.blasNodeOffset < 0 || inst.blasNodeOffset >= (int)t.allBLASNodes.size()) return false;
        t.blasRoots.push_back(t.allBLASNodes[(size_t)inst.blasNodeOffset]);
        // depth of this instance's BLAS: walk it with node indices relative to its offset
        std::vector<BVHNode> sub(t.allBLASNodes.begin() + inst.blasNodeOffset, t.allBLASNodes.end());
        t.maxBLASDepth = std::max(t.maxBLASDepth, subtreeDepth(sub, 0));
    }
    t.tlasDepth = std::max(1, subtreeDepth(t.tlasNodes, 0));
    b = std::move(t);
    return true;
}

}  // namespace rayzen
