/*
 * rz_oracle_bvh.c -- literal C restatement of RayZen's host-side input
 * producers.  TEST INFRASTRUCTURE ONLY; THIS HALF'S PARITY IS UNPINNED (see rz_oracle.h):
 * the reference's BVH.cpp/Mesh.cpp need GLM, which is neither vendored nor
 * installed here, so they cannot be compiled as a cross-check; the only
 * reference outputs available are the node counts/depths the survey recorded
 * (BASELINE.md section 2), which tests/test_bvh.py asserts.
 *
 *   rzo_build_blas   <- RayZen/src/BVH.cpp:11-19 (computeBounds), 22-97
 *                       (findSAHSplit), 99-175 (buildBLAS, SAH branch)
 *   rzo_build_tlas   <- RayZen/src/BVH.cpp:178-240
 *   rzo_world_bounds <- RayZen/src/main.cpp:974-993
 *   rzo_mat4_inverse <- glm::inverse as called at RayZen/src/main.cpp:1001, 1058, 1151 (GLM 0.9.9.8's published algorithm)
 *   rzo_load_obj     <- RayZen/src/Mesh.cpp:6-50
 *
 * Deliberately the reference's O(N log^2 N) algorithm (re-sort at every
 * node): the product's builder (rayzen_amd/csrc/host) is a different,
 * faster formulation that must reproduce these arrays byte for byte.
 */
#include "rz_oracle.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } b3;
static inline b3 B3(float x, float y, float z) { b3 r = {x, y, z}; return r; }
/* glm::min/max(a,b): (b < a) ? b : a  /  (a < b) ? b : a, component-wise */
static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }
static inline b3 min3(b3 a, b3 b) { return B3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static inline b3 max3(b3 a, b3 b) { return B3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
static inline b3 ldv(const float* p) { return B3(p[0], p[1], p[2]); }
static inline float comp(b3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

static inline b3 tri_min(const rzo_triangle* t) { return min3(ldv(t->v0), min3(ldv(t->v1), ldv(t->v2))); }
static inline b3 tri_max(const rzo_triangle* t) { return max3(ldv(t->v0), max3(ldv(t->v1), ldv(t->v2))); }
/* (t.v0 + t.v1 + t.v2) / 3.0f */
static inline b3 centroid(const rzo_triangle* t) {
    b3 s = B3((t->v0[0] + t->v1[0]) + t->v2[0], (t->v0[1] + t->v1[1]) + t->v2[1], (t->v0[2] + t->v1[2]) + t->v2[2]);
    return B3(s.x / 3.0f, s.y / 3.0f, s.z / 3.0f);
}
static inline float half_area2(b3 mn, b3 mx) {   /* 2*(dx*dy + dy*dz + dz*dx) */
    float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return 2.0f * ((dx * dy + dy * dz) + dz * dx);
}

/* BVH.cpp:11-19 */
static void compute_bounds(const rzo_triangle* tris, const int32_t* idx, int start, int end, b3* bmin, b3* bmax) {
    b3 mn = B3(FLT_MAX, FLT_MAX, FLT_MAX), mx = B3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = start; i < end; ++i) {
        const rzo_triangle* t = &tris[idx[i]];
        mn = min3(mn, tri_min(t));
        mx = max3(mx, tri_max(t));
    }
    *bmin = mn; *bmax = mx;
}

typedef struct { float key; int32_t id; } kv_t;
static int kv_cmp(const void* a, const void* b) {       /* std::pair<float,int> operator< */
    const kv_t* p = (const kv_t*)a; const kv_t* q = (const kv_t*)b;
    if (p->key < q->key) return -1;
    if (q->key < p->key) return 1;
    return (p->id < q->id) ? -1 : (p->id > q->id);
}

/* BVH.cpp:22-97.  Returns the split position (count of left items) or -1;
 * sorted_out receives the ids sorted along the best axis. */
static int find_sah_split(const rzo_triangle* tris, const int32_t* idx, int start, int end, int32_t* sorted_out,
                          kv_t* kv, b3* lmin, b3* lmax, b3* rmin, b3* rmax) {
    int bestAxis = -1, bestSplit = -1;
    float bestCost = FLT_MAX;
    int N = end - start;
    if (N <= 4) return -1;
    b3 pmn, pmx;
    compute_bounds(tris, idx, start, end, &pmn, &pmx);
    float parentArea = half_area2(pmn, pmx);
    for (int a = 0; a < 3; ++a) {
        for (int i = 0; i < N; ++i) {
            kv[i].key = comp(centroid(&tris[idx[start + i]]), a);
            kv[i].id = idx[start + i];
        }
        qsort(kv, (size_t)N, sizeof(kv_t), kv_cmp);
        b3 mn = B3(FLT_MAX, FLT_MAX, FLT_MAX), mx = B3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < N; ++i) {
            const rzo_triangle* t = &tris[kv[i].id];
            mn = min3(mn, tri_min(t)); mx = max3(mx, tri_max(t));
            lmin[i] = mn; lmax[i] = mx;
        }
        mn = B3(FLT_MAX, FLT_MAX, FLT_MAX); mx = B3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = N - 1; i >= 0; --i) {
            const rzo_triangle* t = &tris[kv[i].id];
            mn = min3(mn, tri_min(t)); mx = max3(mx, tri_max(t));
            rmin[i] = mn; rmax[i] = mx;
        }
        for (int i = 1; i < N; ++i) {
            float leftArea = half_area2(lmin[i - 1], lmax[i - 1]);
            float rightArea = half_area2(rmin[i], rmax[i]);
            float cost = (leftArea * (float)i + rightArea * (float)(N - i)) / (parentArea + 1e-6f);
            if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = i; }
        }
    }
    if (bestAxis != -1) {
        for (int i = 0; i < N; ++i) {
            kv[i].key = comp(centroid(&tris[idx[start + i]]), bestAxis);
            kv[i].id = idx[start + i];
        }
        qsort(kv, (size_t)N, sizeof(kv_t), kv_cmp);
        for (int i = 0; i < N; ++i) sorted_out[i] = kv[i].id;
    }
    return bestSplit;
}

typedef struct { int nodeIdx, start, end; } entry_t;

static void set_node(rzo_node* n, b3 mn, b3 mx) {
    n->bmin[0] = mn.x; n->bmin[1] = mn.y; n->bmin[2] = mn.z;
    n->bmax[0] = mx.x; n->bmax[1] = mx.y; n->bmax[2] = mx.z;
}

/* BVH.cpp:99-175 (splitMethod == SAH, the default of BVH.h:32) */
int rzo_build_blas(const rzo_triangle* tris, int n, rzo_node* nodes, int32_t* idx) {
    for (int i = 0; i < n; ++i) idx[i] = i;
    int cap = n > 0 ? n : 1;
    entry_t* stack = (entry_t*)malloc(sizeof(entry_t) * (size_t)(2 * cap + 8));
    int32_t* sorted = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)cap);
    b3* lmin = (b3*)malloc(sizeof(b3) * (size_t)cap * 4);
    b3* lmax = lmin + cap; b3* rmin = lmax + cap; b3* rmax = rmin + cap;
    int sp = 0, nn = 0;
    stack[sp].nodeIdx = 0; stack[sp].start = 0; stack[sp].end = n; sp++;
    memset(&nodes[0], 0, sizeof(rzo_node)); nn = 1;
    while (sp > 0) {
        entry_t e = stack[--sp];
        int start = e.start, end = e.end, count = end - start;
        b3 mn, mx;
        compute_bounds(tris, idx, start, end, &mn, &mx);
        set_node(&nodes[e.nodeIdx], mn, mx);
        if (count <= 4) { nodes[e.nodeIdx].leftFirst = start; nodes[e.nodeIdx].count = count; continue; }
        int mid;
        int sah = find_sah_split(tris, idx, start, end, sorted, kv, lmin, lmax, rmin, rmax);
        if (sah > 0 && sah < count) {
            for (int i = 0; i < count; ++i) idx[start + i] = sorted[i];
            mid = start + sah;
        } else {                                              /* BVH.cpp:135-149 */
            int axis = 0;
            float ex = mx.x - mn.x, ey = mx.y - mn.y, ez = mx.z - mn.z;
            if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
            float split = 0.5f * (comp(mn, axis) + comp(mx, axis));
            mid = start;
            for (int i = start; i < end; ++i) {
                if (comp(centroid(&tris[idx[i]]), axis) < split) {
                    int32_t t = idx[i]; idx[i] = idx[mid]; idx[mid] = t; ++mid;
                }
            }
            if (mid == start || mid == end) mid = start + (count / 2);
        }
        int leftIdx = nn, rightIdx = nn + 1;
        nodes[e.nodeIdx].leftFirst = leftIdx; nodes[e.nodeIdx].count = -1;
        memset(&nodes[nn], 0, 2 * sizeof(rzo_node)); nn += 2;
        stack[sp].nodeIdx = rightIdx; stack[sp].start = mid; stack[sp].end = end; sp++;
        stack[sp].nodeIdx = leftIdx; stack[sp].start = start; stack[sp].end = mid; sp++;
    }
    free(stack); free(sorted); free(kv); free(lmin);
    return nn;
}

/* BVH.cpp:178-240 */
int rzo_build_tlas(const rzo_node* roots, int n, rzo_node* nodes, int32_t* idx_out, int* n_idx_out) {
    int cap = n > 0 ? n : 1;
    int32_t* mesh = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    for (int i = 0; i < n; ++i) mesh[i] = i;
    entry_t* stack = (entry_t*)malloc(sizeof(entry_t) * (size_t)(2 * cap + 8));
    int sp = 0, nn = 0, ni = 0;
    stack[sp].nodeIdx = 0; stack[sp].start = 0; stack[sp].end = n; sp++;
    memset(&nodes[0], 0, sizeof(rzo_node)); nn = 1;
    while (sp > 0) {
        entry_t e = stack[--sp];
        int start = e.start, end = e.end, count = end - start;
        b3 mn = B3(FLT_MAX, FLT_MAX, FLT_MAX), mx = B3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = start; i < end; ++i) {
            const rzo_node* r = &roots[mesh[i]];
            mn = min3(mn, ldv(r->bmin)); mx = max3(mx, ldv(r->bmax));
        }
        set_node(&nodes[e.nodeIdx], mn, mx);
        if (count == 1) {
            nodes[e.nodeIdx].leftFirst = ni; nodes[e.nodeIdx].count = 1;
            idx_out[ni++] = mesh[start];
            continue;
        }
        if (count <= 0) {   /* empty scene: BVH.cpp would loop forever on n == 0; emit an empty root */
            nodes[e.nodeIdx].leftFirst = 0; nodes[e.nodeIdx].count = 0;
            continue;
        }
        float ex = mx.x - mn.x, ey = mx.y - mn.y, ez = mx.z - mn.z;
        int axis = 0;
        if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
        float split = 0.5f * (comp(mn, axis) + comp(mx, axis));
        int mid = start;
        for (int i = start; i < end; ++i) {
            const rzo_node* r = &roots[mesh[i]];
            /* (boundsMin + boundsMax) * 0.5f */
            float c = (r->bmin[axis] + r->bmax[axis]) * 0.5f;
            if (c < split) { int32_t t = mesh[i]; mesh[i] = mesh[mid]; mesh[mid] = t; ++mid; }
        }
        if (mid == start || mid == end) mid = start + (count / 2);
        int leftIdx = nn, rightIdx = nn + 1;
        nodes[e.nodeIdx].leftFirst = leftIdx; nodes[e.nodeIdx].count = -1;
        memset(&nodes[nn], 0, 2 * sizeof(rzo_node)); nn += 2;
        stack[sp].nodeIdx = rightIdx; stack[sp].start = mid; stack[sp].end = end; sp++;
        stack[sp].nodeIdx = leftIdx; stack[sp].start = start; stack[sp].end = mid; sp++;
    }
    *n_idx_out = ni;
    free(mesh); free(stack);
    return nn;
}

/* main.cpp:974-993: 8 corners, tc = vec3(transform * vec4(corner, 1)), init +-1e30.
 * glm's mat4 * vec4 (GLM 0.9.9.8, glm/detail/type_mat4x4.inl -- GLM is a third-party, un-vendored, version-unpinned
 * dependency of the reference, RayZen/CMakeLists.txt:16-18, absent from this image; its published algorithm is restated):
 * Mul0 = m[0]*v.x, Mul1 = m[1]*v.y, Add0 = Mul0 + Mul1, Mul2 = m[2]*v.z, Mul3 = m[3]*v.w, Add1 = Mul2 + Mul3,
 * result = Add0 + Add1 -- the column products are added PAIRWISE. */
void rzo_world_bounds(const rzo_node* root, const float m[16], float bmin[3], float bmax[3]) {
    float cx[2] = {root->bmin[0], root->bmax[0]}, cy[2] = {root->bmin[1], root->bmax[1]},
          cz[2] = {root->bmin[2], root->bmax[2]};
    b3 mn = B3(1e30f, 1e30f, 1e30f), mx = B3(-1e30f, -1e30f, -1e30f);
    /* corner order of main.cpp:976-983: x outer, y middle, z inner */
    for (int ix = 0; ix < 2; ++ix) for (int iy = 0; iy < 2; ++iy) for (int iz = 0; iz < 2; ++iz) {
        float x = cx[ix], y = cy[iy], z = cz[iz];
        float t[3];
        for (int r = 0; r < 3; ++r) {
            float add0 = m[r] * x + m[4 + r] * y;
            float add1 = m[8 + r] * z + m[12 + r] * 1.0f;
            t[r] = add0 + add1;
        }
        b3 tc = B3(t[0], t[1], t[2]);
        mn = min3(mn, tc); mx = max3(mx, tc);
    }
    bmin[0] = mn.x; bmin[1] = mn.y; bmin[2] = mn.z;
    bmax[0] = mx.x; bmax[1] = mx.y; bmax[2] = mx.z;
}

/* glm::inverse(mat4) -- main.cpp:1001, 1058, 1151 fill BVHInstance::inverseTransform with it.  GLM 0.9.9.8,
 * glm/detail/func_matrix.inl, compute_inverse<4, 4>, written out scalar by scalar (GLM's m[c][r] is m[4*c + r]):
 * Coef00..23, Fac0..5, Vec0..3, Inv0..3, SignA / SignB, Row0, Dot0, Dot1 = (x + y) + (z + w), times 1 / Dot1. */
void rzo_mat4_inverse(const float m[16], float out[16]) {
#define E(c, r) m[4 * (c) + (r)]
    float Coef00 = E(2,2) * E(3,3) - E(3,2) * E(2,3);
    float Coef02 = E(1,2) * E(3,3) - E(3,2) * E(1,3);
    float Coef03 = E(1,2) * E(2,3) - E(2,2) * E(1,3);
    float Coef04 = E(2,1) * E(3,3) - E(3,1) * E(2,3);
    float Coef06 = E(1,1) * E(3,3) - E(3,1) * E(1,3);
    float Coef07 = E(1,1) * E(2,3) - E(2,1) * E(1,3);
    float Coef08 = E(2,1) * E(3,2) - E(3,1) * E(2,2);
    float Coef10 = E(1,1) * E(3,2) - E(3,1) * E(1,2);
    float Coef11 = E(1,1) * E(2,2) - E(2,1) * E(1,2);
    float Coef12 = E(2,0) * E(3,3) - E(3,0) * E(2,3);
    float Coef14 = E(1,0) * E(3,3) - E(3,0) * E(1,3);
    float Coef15 = E(1,0) * E(2,3) - E(2,0) * E(1,3);
    float Coef16 = E(2,0) * E(3,2) - E(3,0) * E(2,2);
    float Coef18 = E(1,0) * E(3,2) - E(3,0) * E(1,2);
    float Coef19 = E(1,0) * E(2,2) - E(2,0) * E(1,2);
    float Coef20 = E(2,0) * E(3,1) - E(3,0) * E(2,1);
    float Coef22 = E(1,0) * E(3,1) - E(3,0) * E(1,1);
    float Coef23 = E(1,0) * E(2,1) - E(2,0) * E(1,1);
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03}, Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11}, Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19}, Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {E(1,0), E(0,0), E(0,0), E(0,0)}, Vec1[4] = {E(1,1), E(0,1), E(0,1), E(0,1)};
    float Vec2[4] = {E(1,2), E(0,2), E(0,2), E(0,2)}, Vec3[4] = {E(1,3), E(0,3), E(0,3), E(0,3)};
    static const float SignA[4] = {+1.0f, -1.0f, +1.0f, -1.0f}, SignB[4] = {-1.0f, +1.0f, -1.0f, +1.0f};
    float Inverse[16];
    for (int j = 0; j < 4; ++j) {
        float Inv0 = Vec1[j] * Fac0[j] - Vec2[j] * Fac1[j] + Vec3[j] * Fac2[j];
        float Inv1 = Vec0[j] * Fac0[j] - Vec2[j] * Fac3[j] + Vec3[j] * Fac4[j];
        float Inv2 = Vec0[j] * Fac1[j] - Vec1[j] * Fac3[j] + Vec3[j] * Fac5[j];
        float Inv3 = Vec0[j] * Fac2[j] - Vec1[j] * Fac4[j] + Vec2[j] * Fac5[j];
        Inverse[0 + j] = Inv0 * SignA[j];
        Inverse[4 + j] = Inv1 * SignB[j];
        Inverse[8 + j] = Inv2 * SignA[j];
        Inverse[12 + j] = Inv3 * SignB[j];
    }
    float Dot0x = E(0,0) * Inverse[0], Dot0y = E(0,1) * Inverse[4], Dot0z = E(0,2) * Inverse[8], Dot0w = E(0,3) * Inverse[12];
    float Dot1 = (Dot0x + Dot0y) + (Dot0z + Dot0w);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int k = 0; k < 16; ++k) out[k] = Inverse[k] * OneOverDeterminant;
#undef E
}

/* Mesh.cpp:6-50: "v " and "f " lines only; face tokens split at the first '/';
 * 1-based indices; polygons fan-triangulated around the first vertex. */
int rzo_load_obj(const char* path, int materialIndex, rzo_triangle* out, int cap) {
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    size_t vcap = 1024, nv = 0;
    float* verts = (float*)malloc(vcap * 3 * sizeof(float));
    int ntri = 0;
    char* line = NULL; size_t lcap = 0;
    while (getline(&line, &lcap, f) >= 0) {
        if (line[0] == 'v' && line[1] == ' ') {
            float x = 0, y = 0, z = 0;
            sscanf(line + 2, "%f %f %f", &x, &y, &z);
            if (nv == vcap) { vcap *= 2; verts = (float*)realloc(verts, vcap * 3 * sizeof(float)); }
            verts[3 * nv] = x; verts[3 * nv + 1] = y; verts[3 * nv + 2] = z; ++nv;
        } else if (line[0] == 'f' && line[1] == ' ') {
            unsigned int vi[256]; int nvi = 0;
            char* save = NULL;
            for (char* tok = strtok_r(line + 2, " \t\r\n", &save); tok && nvi < 256; tok = strtok_r(NULL, " \t\r\n", &save)) {
                char* slash = strchr(tok, '/');
                if (slash) *slash = 0;
                vi[nvi++] = (unsigned int)atoi(tok);
            }
            if (nvi >= 3) {
                for (int i = 1; i < nvi - 1; ++i) {
                    if (out && ntri < cap) {
                        rzo_triangle* t = &out[ntri];
                        memset(t, 0, sizeof *t);
                        memcpy(t->v0, &verts[3 * (vi[0] - 1)], 12);
                        memcpy(t->v1, &verts[3 * (vi[i] - 1)], 12);
                        memcpy(t->v2, &verts[3 * (vi[i + 1] - 1)], 12);
                        t->materialIndex = materialIndex;
                    }
                    ++ntri;
                }
            }
        }
    }
    free(line); free(verts); fclose(f);
    return ntri;
}
