/*
 * bdpt_bvh.h -- a bounding-volume hierarchy over the triangles, for scenes of many of them: the layout the device
 * walks (bdpt_kernels.hip, closest_hit_bvh) and the host code that builds it at scene upload.
 *
 * The reference meets every ray with every triangle in index order and keeps a hit that is strictly nearer than the
 * best so far (kernel0.cl:434-449): the winner is the hit with the least distance and, among equals, the least index.
 * That is a property of the set of hits, not of the order they are found in, so a search may skip any triangle that
 * cannot be hit and visit the others in any order, as long as it compares (distance, index) -- which is what the
 * hierarchy is for.
 *
 * Which triangles cannot be hit.  The reference's test (kernel0.cl:273-312) accepts a point p = o + t d, computed in
 * float, that passes three edge tests, computed in float: p lies on the ray's line up to the rounding of o + t d,
 * 1.2e-7 (|o| + |p|); it lies in the triangle's plane up to t times the rounding of d.n and the rounding of (a - o).n,
 * some 1e-6 (|o| + |p|), plus theta R for a stored unit normal that is off by the angle theta from the true one (it
 * comes from a cross product of rounded edges; measured per triangle at upload; R the largest distance of a vertex
 * from the centroid); and it lies inside the prism over the triangle along that normal up to the rounding of the edge
 * expressions, some 1e-6 (|p| + R).  A box around the vertices widened by
 *     m   = (0.01 + 10 theta) R + 2e-5 (1 + |c|_1)    per triangle, stored (c the centroid, |.|_1 the sum of absolute coordinates)
 *     q(o) = 2e-5 (1 + |o|_1)                per ray, added to every box by the walk
 * -- ten times the normal's share, some twenty times the rounding's -- therefore contains every point the test can
 * accept, and so does every box that encloses it.  The walk tests the (widened) boxes of a node's two children with
 * the slab method and a relative slack of 1e-5 on the comparisons, a hundred times the rounding of the three
 * operations per slab.
 *
 * Triangles that are not well shaped or whose normal is off by 1e-3 or more (slivers, specks far from the origin:
 * what their edge tests accept need not resemble the triangle) and the
 * few that are far larger than the rest (the walls of a room around a soup of small triangles: their boxes would
 * swallow every node above them) are kept out of the hierarchy in a short list that every ray tests in full.
 *
 * Built top down: a set of triangles is cut where the surface-area heuristic over 16 bins of each axis is least
 * (boxes' areas times triangle counts), provided neither side gets less than an eighth -- else at the median of the
 * longest axis of the centres' bounds -- so the depth stays logarithmic whatever the scene looks like; the binary tree
 * is then folded into nodes of four children.  A walk pushes up to three entries per level on its
 * per-lane stack; the builder reports the most a walk of this tree can need (stack_need), a launch reserves that much
 * LDS per lane, up to BDPT_BVH_STACK entries; a lane whose stack would run over walks again with BDPT_BVH_DEEP entries
 * in its private memory, and if that is not enough either it meets every triangle instead (bdpt_kernels.hip), so any
 * depth is safe.
 */
#ifndef BDPT_BVH_H
#define BDPT_BVH_H

#include "bdpt_layout.h"

#ifndef BDPT_BVH_WIDE
#define BDPT_BVH_WIDE 4           /* children per inner node: 2 or 4 */
#endif
#if BDPT_BVH_WIDE == 2
#define BDPT_BVH_NODE_F4 4        /* (lo0.xyz, hi0.x) (hi0.yz, lo1.xy) (lo1.z, hi1.xyz) (child0, child1, -, -) */
#else
#define BDPT_BVH_NODE_F4 8        /* lo.x, lo.y, lo.z, hi.x, hi.y, hi.z of the four children, their references, padding: one 128-byte line */
#endif
#define BDPT_BVH_EMPTY ((int)0x80000000)   /* reference of a child slot that holds nothing (its box is inside out); no leaf has this value */
#ifndef BDPT_BVH_LEAF_MAX
#define BDPT_BVH_LEAF_MAX 1       /* triangles per leaf */
#endif
#ifndef BDPT_BVH_SAH
#define BDPT_BVH_SAH 16           /* bins per axis of the builder's surface-area split; 0 = median splits only */
#endif
#ifndef BDPT_BVH_SAH_SHARE
#define BDPT_BVH_SAH_SHARE 8      /* a surface-area cut leaves each side at least one part in this many */
#endif
#ifndef BDPT_BVH_WAIT
#define BDPT_BVH_WAIT 8           /* lanes waiting at leaves that end the walk's node loop for their wavefront, when lanes share work */
#endif
#ifndef BDPT_BVH_WAIT_ALONE
#define BDPT_BVH_WAIT_ALONE 4     /* the same when every lane walks for itself (leaf phases are cheaper there) */
#endif
#define BDPT_BVH_SHARE_MIN_TRIANGLES 8192   /* scenes from this size on are walked with the lanes of a wavefront sharing work */
#ifndef BDPT_BVH_STACK
#define BDPT_BVH_STACK 22         /* entries of the walk's per-lane stack in LDS, at most (a launch reserves what its tree can need) */
#endif
#define BDPT_BVH_DEEP 64          /* entries of the stack in a lane's private memory, for the second walk of a lane that ran over */
#define BDPT_BVH_BRUTE_MAX 1024   /* triangles outside the hierarchy, at most */
#define BDPT_BVH_BIG_MAX 64       /* of which oversized ones */
/* a child reference: >= 0 the index of an inner node; < 0 a leaf, ~ref = first entry | (entries - 1) << 28 */
#define BDPT_BVH_NONE 0x7FFFFFFF  /* root of an empty hierarchy */

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

struct bdpt_bvh_host {
    std::vector<float4> nodes;          /* BDPT_BVH_NODE_F4 per inner node */
    std::vector<uint32_t> leaf_tri;     /* triangle indices in leaf order */
    std::vector<uint32_t> brute;        /* triangles outside the hierarchy, ascending */
    int root = BDPT_BVH_NONE;
    int depth = 0;                      /* inner levels of the binary tree above the deepest leaf */
    int stack_need = 0;                 /* entries of the walk's stack that no ray can exceed */
};

/* vertex(i, k) = const float * to the three coordinates of vertex k of triangle i; bound[i] = (centroid, w), w = -inf
 * for a triangle that is not well shaped; big_r2[i] = 1.05 R^2 (infinite for the same ones); theta[i] = |stored normal - true normal|.
 * false: the scene cannot have a hierarchy (too many unbounded triangles, or too deep for the device's stack). */
template <class Vertex>
inline bool bdpt_bvh_build(Vertex vertex, const float4 *bound, const double *big_r2, const double *theta, int n, bdpt_bvh_host &out)
{
    out = bdpt_bvh_host();
    struct Box { float lo[3], hi[3]; };
    std::vector<Box> box((size_t)n);
    std::vector<uint32_t> idx;
    std::vector<double> radii;
    idx.reserve(n);
    auto unbounded = [&](int i) { return !(theta[i] < 1.0e-3) || !(std::isfinite(big_r2[i])) || !(std::isfinite(bound[i].w)) || !std::isfinite(bound[i].x + bound[i].y + bound[i].z); };
    for (int i = 0; i < n; i++)
        if (!unbounded(i)) radii.push_back(std::sqrt(big_r2[i]));
    double median = 0.0;
    if (!radii.empty()) {
        std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
        median = radii[radii.size() / 2];
    }
    /* oversized: more than eight times the median radius; the BDPT_BVH_BIG_MAX largest of them leave the hierarchy */
    std::vector<std::pair<double, uint32_t>> big;
    for (int i = 0; i < n; i++)
        if (!unbounded(i) && std::sqrt(big_r2[i]) > 8.0 * median) big.push_back({std::sqrt(big_r2[i]), (uint32_t)i});
    std::sort(big.begin(), big.end(), [](const std::pair<double, uint32_t> &a, const std::pair<double, uint32_t> &b) { return a.first > b.first; });
    if (big.size() > BDPT_BVH_BIG_MAX) big.resize(BDPT_BVH_BIG_MAX);
    std::vector<char> out_of_tree((size_t)n, 0);
    for (const auto &b : big) out_of_tree[b.second] = 1;
    for (int i = 0; i < n; i++) {
        if (unbounded(i)) out_of_tree[i] = 1;
        if (out_of_tree[i]) { out.brute.push_back((uint32_t)i); continue; }
        const double m = (0.01 + 10.0 * theta[i]) * std::sqrt(big_r2[i] / 1.05) + 2.0e-5 * (1.0 + std::fabs((double)bound[i].x) + std::fabs((double)bound[i].y) + std::fabs((double)bound[i].z));
        for (int k = 0; k < 3; k++) {
            const double v[3] = {vertex(i, 0)[k], vertex(i, 1)[k], vertex(i, 2)[k]};
            box[i].lo[k] = std::nextafter((float)(std::min(v[0], std::min(v[1], v[2])) - m), -INFINITY);
            box[i].hi[k] = std::nextafter((float)(std::max(v[0], std::max(v[1], v[2])) + m), INFINITY);
        }
        idx.push_back((uint32_t)i);
    }
    if (out.brute.size() > BDPT_BVH_BRUTE_MAX) return false;
    if (idx.empty()) return true;
    out.leaf_tri.reserve(idx.size());
    struct Pair { Box b[2]; int ref[2]; };
    std::vector<Pair> pairs;                               /* the binary tree */
    pairs.reserve(idx.size());

    struct Ref { int ref; Box b; int depth; };
    /* recursion on [first, last) of idx; depth-first, so a subtree's nodes and leaves are contiguous */
    struct Builder {
        const float4 *bound;
        std::vector<Box> &box;
        std::vector<uint32_t> &idx;
        bdpt_bvh_host &out;
        std::vector<Pair> &pairs;
        Ref build(size_t first, size_t last, int level = 0)
        {
            Ref r;
            for (int k = 0; k < 3; k++) { r.b.lo[k] = INFINITY; r.b.hi[k] = -INFINITY; }
            const size_t count = last - first;
            if (count <= BDPT_BVH_LEAF_MAX) {
                std::sort(idx.begin() + first, idx.begin() + last);
                const size_t at = out.leaf_tri.size();
                for (size_t j = first; j < last; j++) {
                    const uint32_t i = idx[j];
                    out.leaf_tri.push_back(i);
                    for (int k = 0; k < 3; k++) { r.b.lo[k] = std::min(r.b.lo[k], box[i].lo[k]); r.b.hi[k] = std::max(r.b.hi[k], box[i].hi[k]); }
                }
                r.ref = ~(int)((uint32_t)at | (uint32_t)(count - 1) << 28);
                r.depth = 0;
                return r;
            }
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (size_t j = first; j < last; j++) {
                const float4 c = bound[idx[j]];
                const float cc[3] = {c.x, c.y, c.z};
                for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], cc[k]); chi[k] = std::max(chi[k], cc[k]); }
            }
            int axis = 0;
            for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
            size_t mid = first + count / 2;
            auto centre = [&](uint32_t t) { return axis == 0 ? bound[t].x : axis == 1 ? bound[t].y : bound[t].z; };
            bool split = false;
#if BDPT_BVH_SAH > 0
            /* surface-area heuristic over BDPT_BVH_SAH bins of each axis: the cut that minimises the boxes' areas times
             * their triangle counts; taken when neither side gets less than an eighth (depth stays logarithmic), else
             * the median of the longest axis below */
            if (count > 8 && level < 64) {
                constexpr int NB = BDPT_BVH_SAH;
                auto area = [](const Box &x) {
                    const double dx = (double)x.hi[0] - x.lo[0], dy = (double)x.hi[1] - x.lo[1], dz = (double)x.hi[2] - x.lo[2];
                    return dx * dy + dy * dz + dz * dx;
                };
                double best = INFINITY;
                int cut = -1, cut_axis = -1;
                for (int ax = 0; ax < 3; ax++) {
                    if (!(chi[ax] > clo[ax])) continue;
                    Box bb[NB];
                    size_t bn[NB];
                    for (int q = 0; q < NB; q++) { bn[q] = 0; for (int k = 0; k < 3; k++) { bb[q].lo[k] = INFINITY; bb[q].hi[k] = -INFINITY; } }
                    const double scale = NB / ((double)chi[ax] - clo[ax]);
                    for (size_t j = first; j < last; j++) {
                        const uint32_t t = idx[j];
                        const float c = ax == 0 ? bound[t].x : ax == 1 ? bound[t].y : bound[t].z;
                        const int q = std::min(NB - 1, std::max(0, (int)((c - clo[ax]) * scale)));
                        bn[q]++;
                        for (int k = 0; k < 3; k++) { bb[q].lo[k] = std::min(bb[q].lo[k], box[t].lo[k]); bb[q].hi[k] = std::max(bb[q].hi[k], box[t].hi[k]); }
                    }
                    double right_area[NB];
                    size_t right_n[NB];
                    Box acc;
                    for (int k = 0; k < 3; k++) { acc.lo[k] = INFINITY; acc.hi[k] = -INFINITY; }
                    size_t n = 0;
                    for (int q = NB - 1; q > 0; q--) {
                        n += bn[q];
                        for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], bb[q].lo[k]); acc.hi[k] = std::max(acc.hi[k], bb[q].hi[k]); }
                        right_area[q] = n ? area(acc) : 0.0;
                        right_n[q] = n;
                    }
                    for (int k = 0; k < 3; k++) { acc.lo[k] = INFINITY; acc.hi[k] = -INFINITY; }
                    n = 0;
                    for (int q = 0; q + 1 < NB; q++) {
                        n += bn[q];
                        for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], bb[q].lo[k]); acc.hi[k] = std::max(acc.hi[k], bb[q].hi[k]); }
                        if (n < count / BDPT_BVH_SAH_SHARE || right_n[q + 1] < count / BDPT_BVH_SAH_SHARE) continue;
                        const double cost = area(acc) * (double)n + right_area[q + 1] * (double)right_n[q + 1];
                        if (cost < best) { best = cost; cut = q; cut_axis = ax; }
                    }
                }
                if (cut >= 0) {
                    const double scale = NB / ((double)chi[cut_axis] - clo[cut_axis]);
                    const auto m = std::partition(idx.begin() + first, idx.begin() + last, [&](uint32_t t) {
                        const float c = cut_axis == 0 ? bound[t].x : cut_axis == 1 ? bound[t].y : bound[t].z;
                        return std::min(NB - 1, std::max(0, (int)((c - clo[cut_axis]) * scale))) <= cut;
                    });
                    mid = (size_t)(m - idx.begin());
                    split = mid > first && mid < last;
                    if (!split) mid = first + count / 2;
                }
            }
#endif
            if (!split)
                std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + last, [&](uint32_t a, uint32_t b) {
                    const float ca = centre(a), cb = centre(b);
                    return ca < cb || (ca == cb && a < b);
                });
            const size_t node = pairs.size();
            pairs.emplace_back();
            const Ref a = build(first, mid, level + 1), b = build(mid, last, level + 1);
            pairs[node].b[0] = a.b; pairs[node].b[1] = b.b;
            pairs[node].ref[0] = a.ref; pairs[node].ref[1] = b.ref;
            for (int k = 0; k < 3; k++) { r.b.lo[k] = std::min(a.b.lo[k], b.b.lo[k]); r.b.hi[k] = std::max(a.b.hi[k], b.b.hi[k]); }
            r.ref = (int)node;
            r.depth = 1 + std::max(a.depth, b.depth);
            return r;
        }
    } builder{bound, box, idx, out, pairs};
    if (idx.size() >= ((size_t)1 << 28)) return false;
    const Ref root = builder.build(0, idx.size());
    out.depth = root.depth;
    static_assert(sizeof(float) == sizeof(int), "child references travel as the bits of a float");
    auto as_float = [](int ref) { float f; std::memcpy(&f, &ref, 4); return f; };
#if BDPT_BVH_WIDE == 2
    out.root = root.ref;
    out.nodes.resize(pairs.size() * BDPT_BVH_NODE_F4);
    for (size_t n = 0; n < pairs.size(); n++) {
        const Box &a = pairs[n].b[0], &b = pairs[n].b[1];
        float4 *q = &out.nodes[n * BDPT_BVH_NODE_F4];
        q[0] = make_float4(a.lo[0], a.lo[1], a.lo[2], a.hi[0]);
        q[1] = make_float4(a.hi[1], a.hi[2], b.lo[0], b.lo[1]);
        q[2] = make_float4(b.lo[2], b.hi[0], b.hi[1], b.hi[2]);
        q[3] = make_float4(as_float(pairs[n].ref[0]), as_float(pairs[n].ref[1]), 0.0f, 0.0f);
    }
    out.stack_need = root.depth;
    return root.depth <= BDPT_BVH_STACK;
#else
    /* four children per node: a node's children are replaced by their own children, the one with the largest box
     * first, until there are four or only leaves are left */
    if (root.ref < 0) { out.root = root.ref; out.stack_need = 1; return true; }
    struct Wide {
        const std::vector<Pair> &pairs;
        bdpt_bvh_host &out;
        float (*as_float)(int);
        static double area(const Box &b) {
            const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
            return x * y + y * z + z * x;
        }
        int collapse(int pair, int &need)
        {
            Box b[4];
            int ref[4], n = 2;
            b[0] = pairs[pair].b[0]; b[1] = pairs[pair].b[1];
            ref[0] = pairs[pair].ref[0]; ref[1] = pairs[pair].ref[1];
            while (n < 4) {
                int pick = -1;
                for (int k = 0; k < n; k++)
                    if (ref[k] >= 0 && (pick < 0 || area(b[k]) > area(b[pick]))) pick = k;
                if (pick < 0) break;
                const Pair &p = pairs[ref[pick]];
                b[pick] = p.b[0]; ref[pick] = p.ref[0];
                b[n] = p.b[1]; ref[n] = p.ref[1];
                n++;
            }
            const size_t node = out.nodes.size() / BDPT_BVH_NODE_F4;
            out.nodes.resize(out.nodes.size() + BDPT_BVH_NODE_F4);
            for (int k = n; k < 4; k++) {
                for (int a = 0; a < 3; a++) { b[k].lo[a] = INFINITY; b[k].hi[a] = -INFINITY; }
                ref[k] = BDPT_BVH_EMPTY;
            }
            int below = 0;
            for (int k = 0; k < n; k++)
                if (ref[k] >= 0) { int sub = 0; ref[k] = collapse(ref[k], sub); below = std::max(below, sub); }
            need = (n - 1) + below;                          /* the other children wait on the stack while one is walked */
            float4 *q = &out.nodes[node * BDPT_BVH_NODE_F4];
            for (int a = 0; a < 3; a++) {
                q[a] = make_float4(b[0].lo[a], b[1].lo[a], b[2].lo[a], b[3].lo[a]);
                q[3 + a] = make_float4(b[0].hi[a], b[1].hi[a], b[2].hi[a], b[3].hi[a]);
            }
            q[6] = make_float4(as_float(ref[0]), as_float(ref[1]), as_float(ref[2]), as_float(ref[3]));
            q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            return (int)node;
        }
    } wide{pairs, out, +as_float};
    out.nodes.reserve(pairs.size() / 2 * BDPT_BVH_NODE_F4);
    out.root = wide.collapse(root.ref, out.stack_need);
    return true;                                             /* a walk whose stack runs full finishes without the hierarchy */
#endif
}
#endif
