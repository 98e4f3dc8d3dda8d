// BVH.cpp -- host BVH construction for the render library's inputs.
//
// Produces, byte for byte, the node and index arrays of RayZen's builder
// (RayZen/src/BVH.cpp:99-175 buildBLAS with the full-sweep SAH of :22-97;
// :178-240 buildTLAS), but is organised differently: the reference re-sorts
// every node's triangles three times (O(N log^2 N)); here each axis is
// sorted ONCE over the whole mesh by the reference's key (centroid[axis],
// triangle id) -- a total order, so any subset's sorted order is the global
// order restricted to it -- and the three lists are partitioned stably at
// every split (O(N log N)).  The sweep arithmetic (prefix/suffix boxes,
// cost = (A_l*i + A_r*(N-i)) / (A_parent + 1e-6f), first strict minimum over
// axis-major then split-minor order) is the reference's, operation for
// operation, so the chosen splits -- and with them node numbering, leaf
// ranges and index order -- are identical.
#include "RayZenScene.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <limits>
#include <numeric>

namespace rayzen {

namespace {

struct Box { vec3 mn, mx; };

inline float surface(const vec3& mn, const vec3& mx) {
    float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
inline vec3 triMin(const Triangle& t) { return vmin(t.v0, vmin(t.v1, t.v2)); }
inline vec3 triMax(const Triangle& t) { return vmax(t.v0, vmax(t.v1, t.v2)); }

struct BuildEntry { int nodeIdx, start, end; };

}  // namespace

void BVH::buildBLAS(const std::vector<Triangle>& tris) { buildBLAS(tris.data(), (int)tris.size()); }

void BVH::buildBLAS(const Triangle* tris, int n) {
    const float FMAX = std::numeric_limits<float>::max();
    triIndices.resize(n);
    std::iota(triIndices.begin(), triIndices.end(), 0);
    nodes.clear();
    nodes.reserve((size_t)n * 2 + 1);

    // per-triangle boxes and centroids, computed once with the reference's expressions
    std::vector<vec3> tmin(n), tmax(n), cen(n);
    for (int i = 0; i < n; ++i) {
        tmin[i] = triMin(tris[i]);
        tmax[i] = triMax(tris[i]);
        cen[i] = (tris[i].v0 + tris[i].v1 + tris[i].v2) / 3.0f;
    }
    // three global orders by (centroid[a], id)
    std::vector<int> order[3];
    if (splitMethod == BVHSplitMethod::SAH) {
        for (int a = 0; a < 3; ++a) {
            order[a].resize(n);
            std::iota(order[a].begin(), order[a].end(), 0);
            std::sort(order[a].begin(), order[a].end(), [&](int p, int q) {
                float kp = cen[p][a], kq = cen[q][a];
                if (kp < kq) return true;
                if (kq < kp) return false;
                return p < q;
            });
        }
    }
    std::vector<uint8_t> goesLeft(n, 0);
    std::vector<int> scratch(n);
    std::vector<vec3> rMin(n), rMax(n);

    std::vector<BuildEntry> stack;
    stack.push_back({0, 0, n});
    nodes.push_back(BVHNode{});
    while (!stack.empty()) {
        BuildEntry e = stack.back();
        stack.pop_back();
        const int start = e.start, end = e.end, count = end - start;
        vec3 bmin(FMAX), bmax(-FMAX);
        for (int i = start; i < end; ++i) {
            bmin = vmin(bmin, tmin[triIndices[i]]);
            bmax = vmax(bmax, tmax[triIndices[i]]);
        }
        nodes[e.nodeIdx].boundsMin = bmin;
        nodes[e.nodeIdx].boundsMax = bmax;
        if (count <= 4) {
            nodes[e.nodeIdx].leftFirst = start;
            nodes[e.nodeIdx].count = count;
            continue;
        }
        int mid = start;
        bool usedSAH = false;
        if (splitMethod == BVHSplitMethod::SAH) {
            const int N = count;
            const float parentArea = surface(bmin, bmax);
            int bestAxis = -1, bestSplit = -1;
            float bestCost = FMAX;
            for (int a = 0; a < 3; ++a) {
                const int* ord = order[a].data() + start;
                vec3 mn(FMAX), mx(-FMAX);
                for (int i = N - 1; i >= 0; --i) {
                    mn = vmin(mn, tmin[ord[i]]);
                    mx = vmax(mx, tmax[ord[i]]);
                    rMin[i] = mn; rMax[i] = mx;
                }
                mn = vec3(FMAX); mx = vec3(-FMAX);
                for (int i = 1; i < N; ++i) {
                    mn = vmin(mn, tmin[ord[i - 1]]);
                    mx = vmax(mx, tmax[ord[i - 1]]);
                    float leftArea = surface(mn, mx);
                    float rightArea = surface(rMin[i], rMax[i]);
                    float cost = (leftArea * (float)i + rightArea * (float)(N - i)) / (parentArea + 1e-6f);
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = i; }
                }
            }
            if (bestAxis != -1 && bestSplit > 0 && bestSplit < N) {
                usedSAH = true;
                const int* ord = order[bestAxis].data() + start;
                for (int i = 0; i < N; ++i) {
                    triIndices[start + i] = ord[i];
                    goesLeft[ord[i]] = (uint8_t)(i < bestSplit);
                }
                mid = start + bestSplit;
            }
        }
        if (!usedSAH) {   // midpoint split: BVH.cpp:135-149 (SAH fallback) == :151-164 (Midpoint method)
            int axis = 0;
            vec3 extent = bmax - bmin;
            if (extent.y > extent.x && extent.y > extent.z) axis = 1;
            else if (extent.z > extent.x) axis = 2;
            float split = 0.5f * (bmin[axis] + bmax[axis]);
            mid = start;
            for (int i = start; i < end; ++i) {
                if (cen[triIndices[i]][axis] < split) {
                    std::swap(triIndices[i], triIndices[mid]);
                    ++mid;
                }
            }
            if (mid == start || mid == end) mid = start + (count / 2);
            for (int i = start; i < end; ++i) goesLeft[triIndices[i]] = (uint8_t)(i < mid);
        }
        if (splitMethod == BVHSplitMethod::SAH) {
            // stable partition of the three sorted lists by side
            for (int a = 0; a < 3; ++a) {
                int* ord = order[a].data() + start;
                int l = 0, r = 0;
                for (int i = 0; i < count; ++i) {
                    if (goesLeft[ord[i]]) ord[l++] = ord[i];
                    else scratch[r++] = ord[i];
                }
                std::copy(scratch.begin(), scratch.begin() + r, ord + l);
            }
        }
        int leftIdx = (int)nodes.size();
        int rightIdx = leftIdx + 1;
        nodes[e.nodeIdx].leftFirst = leftIdx;
        nodes[e.nodeIdx].count = -1;
        nodes.push_back(BVHNode{});
        nodes.push_back(BVHNode{});
        stack.push_back({rightIdx, mid, end});
        stack.push_back({leftIdx, start, mid});
    }
}

// BVH::buildTLAS (RayZen/src/BVH.cpp:178-240): midpoint split on the longest axis of the instances' world boxes, ONE instance
// per leaf.  The output -- node numbering included -- is a function of the instance ranges alone, which is how it is built
// here (and, level by level, on the device: rz_tlas_device.hip):
//   * a range of c instances becomes a subtree of exactly 2c - 1 nodes;
//   * the reference appends a node's two children as a pair when it visits the node, and visits in depth-first order, left
//     subtree first: the k-th INTERNAL node it visits (k from 0, the root) therefore owns nodes 2k + 1 and 2k + 2; the left
//     child of internal node k is internal node k + 1 (if it has more than one instance), the right child internal node
//     k + (instances on the left) -- a left subtree over m instances holds m - 1 internal nodes;
//   * leaves are reached left to right, so the leaf over position p of the (partitioned) order is the p-th one written to
//     the index array: leftFirst = p, and the index array IS the final order.
// So the nodes live in an array sized up front, each range writes its own node and names its children by arithmetic, and
// the ranges may be worked off in any order (a plain work list here).  Per node the arithmetic is the reference's: the
// box folded over the range in order (glm::min / glm::max keep the first of equals, so a zero keeps its sign), the split
// plane at the middle of the longest extent (x on ties, as BVH.cpp:196-198), the centre test `c < split` and the swap
// partition of BVH.cpp:201-209 -- its exact permutation is part of the output -- and the fall-back split in the middle.
void BVH::buildTLAS(const std::vector<BVHInstance>& meshInstances, const std::vector<BVHNode>& meshRootNodes) {
    const float FMAX = std::numeric_limits<float>::max();
    instances = meshInstances;
    const int n = (int)meshInstances.size();
    triIndices.resize((size_t)n);
    std::iota(triIndices.begin(), triIndices.end(), 0);         // the order being partitioned; final = the index array
    nodes.assign((size_t)std::max(1, 2 * n - 1), BVHNode{});
    if (n <= 0) {           // empty scene (the reference would not terminate): an empty root with the fold's identity box
        nodes[0].boundsMin = vec3(FMAX); nodes[0].boundsMax = vec3(-FMAX);
        nodes[0].leftFirst = 0; nodes[0].count = 0;
        return;
    }
    struct Range { int node, internalRank, first, last; };      // instances [first, last) of the order
    std::vector<Range> work{{0, 0, 0, n}};
    while (!work.empty()) {
        const Range r = work.back();
        work.pop_back();
        BVHNode& N = nodes[(size_t)r.node];
        vec3 lo(FMAX), hi(-FMAX);
        for (int p = r.first; p < r.last; ++p) {
            const BVHNode& box = meshRootNodes[(size_t)triIndices[(size_t)p]];
            lo = vmin(lo, box.boundsMin);
            hi = vmax(hi, box.boundsMax);
        }
        N.boundsMin = lo;
        N.boundsMax = hi;
        const int members = r.last - r.first;
        if (members == 1) { N.leftFirst = r.first; N.count = 1; continue; }
        const vec3 size = hi - lo;
        const int axis = (size.y > size.x && size.y > size.z) ? 1 : (size.z > size.x ? 2 : 0);
        const float plane = 0.5f * (lo[axis] + hi[axis]);
        int cut = r.first;                                      // BVH.cpp:201-209: the instances whose centre lies below the plane, swapped to the front
        for (int p = r.first; p < r.last; ++p) {
            const BVHNode& box = meshRootNodes[(size_t)triIndices[(size_t)p]];
            if ((box.boundsMin[axis] + box.boundsMax[axis]) * 0.5f < plane) std::swap(triIndices[(size_t)p], triIndices[(size_t)cut++]);
        }
        if (cut == r.first || cut == r.last) cut = r.first + members / 2;
        N.leftFirst = 2 * r.internalRank + 1;
        N.count = -1;
        work.push_back({N.leftFirst, r.internalRank + 1, r.first, cut});
        work.push_back({N.leftFirst + 1, r.internalRank + (cut - r.first), cut, r.last});
    }
}

int BVH::depth() const {
    if (nodes.empty()) return 0;
    int best = 1;
    std::vector<std::pair<int, int>> st;
    st.push_back({0, 1});
    while (!st.empty()) {
        auto [n, d] = st.back();
        st.pop_back();
        best = std::max(best, d);
        if (nodes[n].count < 0) {
            st.push_back({nodes[n].leftFirst, d + 1});
            st.push_back({nodes[n].leftFirst + 1, d + 1});
        }
    }
    return best;
}

}  // namespace rayzen
