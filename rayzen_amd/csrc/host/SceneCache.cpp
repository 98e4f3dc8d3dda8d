// SceneCache.cpp -- see SceneCache.h (RayZen/src/main.cpp:94-133, 914-939, 1037-1043).
#include "SceneCache.h"

#include <algorithm>
#include <map>
#include <sys/stat.h>
#include <sys/types.h>

namespace rayzen {

namespace {
std::string join(const std::string& dir, const char* name) {
    if (dir.empty()) return name;
    return dir.back() == '/' ? dir + name : dir + "/" + name;
}
// Depth of the tree whose root is nodes[base] and whose child indices are relative to `base` (a BLAS inside the
// concatenated array, or the TLAS with base 0): walked in place.
int subtreeDepth(const std::vector<BVHNode>& nodes, size_t base) {
    if (base >= nodes.size()) return 0;
    const size_t span = nodes.size() - base;
    int best = 1;
    std::vector<std::pair<int, int>> st{{0, 1}};
    size_t guard = 0;
    while (!st.empty() && guard++ <= span * 2) {
        auto [n, d] = st.back();
        st.pop_back();
        best = std::max(best, d);
        const BVHNode& N = nodes[base + (size_t)n];
        if (N.count < 0 && N.leftFirst >= 0 && (size_t)N.leftFirst + 1 < span) {
            st.push_back({N.leftFirst, d + 1});
            st.push_back({N.leftFirst + 1, d + 1});
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
    std::map<int, int> depthOf;
    for (const BVHInstance& inst : t.meshInstances) {
        if (inst.blasNodeOffset < 0 || inst.blasNodeOffset >= (int)t.allBLASNodes.size()) return false;
        t.blasRoots.push_back(t.allBLASNodes[(size_t)inst.blasNodeOffset]);
        // depth of this instance's BLAS, once per distinct BLAS (instances of one mesh share an offset)
        auto seen = depthOf.find(inst.blasNodeOffset);
        if (seen == depthOf.end()) seen = depthOf.emplace(inst.blasNodeOffset, subtreeDepth(t.allBLASNodes, (size_t)inst.blasNodeOffset)).first;
        t.maxBLASDepth = std::max(t.maxBLASDepth, seen->second);
    }
    t.tlasDepth = std::max(1, subtreeDepth(t.tlasNodes, (size_t)0));
    b = std::move(t);
    return true;
}

bool saveBVHToFile(const std::string& base, const BVH& bvh) {
    return saveVectorToFile(base + ".nodes.bin", bvh.nodes) && saveVectorToFile(base + ".tris.bin", bvh.triIndices);
}
bool loadBVHFromFile(const std::string& base, BVH& bvh) {
    return loadVectorFromFile(base + ".nodes.bin", bvh.nodes) && loadVectorFromFile(base + ".tris.bin", bvh.triIndices);
}

namespace {
bool exists(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0;
}
void makeDirs(const std::string& dir) {          // fs::create_directories
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && !exists(cur)) ::mkdir(cur.c_str(), 0777);
        }
        if (i < dir.size()) cur.push_back(dir[i]);
    }
}
void finishDerived(SceneBuffers& b) {            // what the renderer wants besides the six arrays
    b.blasRoots.clear();
    b.maxBLASDepth = 1;
    std::map<int, int> depthOf;
    for (const BVHInstance& inst : b.meshInstances) {
        if (inst.blasNodeOffset < 0 || inst.blasNodeOffset >= (int)b.allBLASNodes.size()) { b.blasRoots.push_back(BVHNode{}); continue; }
        b.blasRoots.push_back(b.allBLASNodes[(size_t)inst.blasNodeOffset]);
        auto seen = depthOf.find(inst.blasNodeOffset);
        if (seen == depthOf.end()) seen = depthOf.emplace(inst.blasNodeOffset, subtreeDepth(b.allBLASNodes, (size_t)inst.blasNodeOffset)).first;
        b.maxBLASDepth = std::max(b.maxBLASDepth, seen->second);
    }
    b.tlasDepth = std::max(1, subtreeDepth(b.tlasNodes, (size_t)0));
}
}  // namespace

bool initializeSSBOsCached(const Scene& scene, const std::string& dirIn, bool forceRebuildBVH, SceneBuffers& out,
                           CacheReport* report) {
    CacheReport rep;
    const std::string cacheDir = (dirIn.empty() || dirIn.back() == '/') ? dirIn : dirIn + "/";
    if (!cacheDir.empty() && !exists(cacheDir)) makeDirs(cacheDir);                       // main.cpp:899-903

    SceneBuffers b;
    b.blasBuilder = out.blasBuilder;
    const std::string ssbo = cacheDir + "ssbo_v2_";
    bool loadedSSBOCache = false;
    if (!forceRebuildBVH && exists(ssbo + "triangles.bin") && exists(ssbo + "blasnodes.bin") && exists(ssbo + "blastris.bin") &&
        exists(ssbo + "instances.bin") && exists(ssbo + "tlasnodes.bin") && exists(ssbo + "tlastris.bin")) {   // main.cpp:914-920
        loadedSSBOCache = loadVectorFromFile(ssbo + "triangles.bin", b.allTriangles) &&
                          loadVectorFromFile(ssbo + "blasnodes.bin", b.allBLASNodes) &&
                          loadVectorFromFile(ssbo + "blastris.bin", b.allBLASTriIndices) &&
                          loadVectorFromFile(ssbo + "instances.bin", b.meshInstances) &&
                          loadVectorFromFile(ssbo + "tlasnodes.bin", b.tlasNodes) &&
                          loadVectorFromFile(ssbo + "tlastris.bin", b.tlasTriIndices);
        if (loadedSSBOCache && b.meshInstances.size() != scene.gameObjects.size()) {      // main.cpp:929-934
            rep.ssboInvalidated = true;
            loadedSSBOCache = false;
        }
        if (!loadedSSBOCache) {
            b.allTriangles.clear(); b.allBLASNodes.clear(); b.allBLASTriIndices.clear();
            b.meshInstances.clear(); b.tlasNodes.clear(); b.tlasTriIndices.clear();
        }
    }

    if (!loadedSSBOCache) {
        static const Mesh kEmpty;
        std::vector<BVH> meshBLAS(scene.gameObjects.size());
        std::vector<BVHNode> worldRootNodes;
        worldRootNodes.reserve(scene.gameObjects.size());
        int nodeOffset = 0, triOffset = 0;
        bool loadedAllBLAS = true;
        for (size_t i = 0; i < scene.gameObjects.size(); ++i) {                            // main.cpp:951-1007
            const GameObject& obj = scene.gameObjects[i];
            const Mesh& mesh = obj.mesh ? *obj.mesh : kEmpty;
            const std::string blasBase = cacheDir + "mesh" + std::to_string(i);
            bool loaded = false;
            if (!forceRebuildBVH && exists(blasBase + ".nodes.bin") && exists(blasBase + ".tris.bin"))
                loaded = loadBVHFromFile(blasBase, meshBLAS[i]) && !meshBLAS[i].nodes.empty();
            if (!loaded) {
                meshBLAS[i] = BVH{};
                if (b.blasBuilder) { if (!b.blasBuilder(mesh, meshBLAS[i])) return false; }
                else meshBLAS[i].buildBLAS(mesh.triangles);
                saveBVHToFile(blasBase, meshBLAS[i]);
                ++rep.blasBuilt;
            } else {
                ++rep.blasLoaded;
            }
            const size_t triBase = b.allTriangles.size();
            b.allTriangles.insert(b.allTriangles.end(), mesh.triangles.begin(), mesh.triangles.end());
            worldRootNodes.push_back(worldRootNode(meshBLAS[i].nodes[0], obj.transform));   // main.cpp:974-993
            BVHInstance inst;
            inst.blasNodeOffset = nodeOffset;
            inst.blasTriOffset = triOffset;
            inst.globalTriOffset = (int)triBase;
            inst.meshIndex = (int)i;
            inst.transform = obj.transform;
            inst.inverseTransform = inverse(obj.transform);
            b.meshInstances.push_back(inst);
            nodeOffset += (int)meshBLAS[i].nodes.size();
            triOffset += (int)meshBLAS[i].triIndices.size();
            loadedAllBLAS = loadedAllBLAS && loaded;
        }
        BVH tlas;
        const std::string tlasBase = cacheDir + "scene_tlas";
        bool loadedTLAS = false;
        if (!forceRebuildBVH && loadedAllBLAS && exists(tlasBase + ".nodes.bin") && exists(tlasBase + ".tris.bin") &&
            exists(cacheDir + "instances.bin")) {                                           // main.cpp:1012-1018
            std::vector<BVHInstance> cached;
            loadedTLAS = loadBVHFromFile(tlasBase, tlas) && loadVectorFromFile(cacheDir + "instances.bin", cached);
            if (loadedTLAS) b.meshInstances = cached;      // the reference overwrites the records it has just assembled
        }
        if (!loadedTLAS) {
            tlas = BVH{};
            tlas.buildTLAS(b.meshInstances, worldRootNodes);
            saveBVHToFile(tlasBase, tlas);
            saveVectorToFile(cacheDir + "instances.bin", b.meshInstances);
        }
        rep.tlasLoaded = loadedTLAS;
        for (const BVH& blas : meshBLAS) {                                                  // main.cpp:1030-1035
            b.allBLASNodes.insert(b.allBLASNodes.end(), blas.nodes.begin(), blas.nodes.end());
            b.allBLASTriIndices.insert(b.allBLASTriIndices.end(), blas.triIndices.begin(), blas.triIndices.end());
        }
        b.tlasNodes = tlas.nodes;
        b.tlasTriIndices = tlas.triIndices;
        saveSceneCache(cacheDir, b);                                                        // main.cpp:1037-1043
    } else {
        rep.ssboLoaded = true;
        for (size_t i = 0; i < b.meshInstances.size() && i < scene.gameObjects.size(); ++i) {   // main.cpp:1054-1060
            b.meshInstances[i].transform = scene.gameObjects[i].transform;
            b.meshInstances[i].meshIndex = (int)i;
            b.meshInstances[i].inverseTransform = inverse(scene.gameObjects[i].transform);
        }
    }
    finishDerived(b);
    out = std::move(b);
    if (report) *report = rep;
    return true;
}

}  // namespace rayzen
