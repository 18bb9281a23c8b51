// Binned-SAH BVH2 builder (host). Leaves hold <= 2 or 3 triangles (BLAS, by mesh size; the walk takes up to 4) or exactly
// one instance (TLAS). Children of an inner node are adjacent so the walk
// fetches both with one 64-B access. Node boxes are padded by a few ulps so the
// conservative slab test can never reject a point the watertight triangle test
// accepts.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace thx {

namespace {

constexpr uint32_t kSmallMeshTriangles = 1u << 20;

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float p[3]) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float cen[3];
    uint32_t id;
};

struct Builder {
    std::vector<Prim>& prims;
    std::vector<BvhNode>& nodes;
    uint32_t leafMax;
    bool median;
    uint32_t maxDepth = 0;

    static void store(BvhNode& n, const Box& b) {
        for (int k = 0; k < 3; ++k) {
            float pad = 2e-6f * std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])) + 1e-30f;
            n.lo[k] = b.lo[k] - pad;
            n.hi[k] = b.hi[k] + pad;
        }
    }

    void build(uint32_t node, uint32_t first, uint32_t count, uint32_t depth) {
        maxDepth = std::max(maxDepth, depth);
        Box box, cbox;
        box.reset();
        cbox.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            box.grow(prims[i].box);
            cbox.grow(prims[i].cen);
        }
        store(nodes[node], box);
        if (count <= leafMax) {
            nodes[node].leftFirst = first;
            nodes[node].count = count;
            return;
        }
        int axis = 0;
        float ext[3] = {cbox.hi[0] - cbox.lo[0], cbox.hi[1] - cbox.lo[1], cbox.hi[2] - cbox.lo[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        uint32_t mid = first + count / 2;
        bool done = false;
        if (!median && ext[axis] > 0.0f) {
            // binned SAH over all three axes
            constexpr int NB = 16;
            float best = std::numeric_limits<float>::infinity();
            int bestAxis = -1, bestBin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                if (!(ext[ax] > 0.0f)) continue;
                Box bb[NB];
                uint32_t bc[NB];
                for (int b = 0; b < NB; ++b) { bb[b].reset(); bc[b] = 0; }
                float scale = NB / ext[ax];
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = std::min(NB - 1, std::max(0, (int)((prims[i].cen[ax] - cbox.lo[ax]) * scale)));
                    bb[b].grow(prims[i].box);
                    bc[b]++;
                }
                float rightArea[NB];
                uint32_t rightCount[NB];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = NB - 1; b > 0; --b) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    rightArea[b] = acc.area();
                    rightCount[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    if (cnt == 0 || rightCount[b + 1] == 0) continue;
                    float cost = acc.area() * cnt + rightArea[b + 1] * rightCount[b + 1];
                    if (cost < best) { best = cost; bestAxis = ax; bestBin = b; }
                }
            }
            if (bestAxis >= 0) {
                float scale = NB / ext[bestAxis];
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim& p) {
                    int b = std::min(NB - 1, std::max(0, (int)((p.cen[bestAxis] - cbox.lo[bestAxis]) * scale)));
                    return b <= bestBin;
                });
                mid = (uint32_t)(it - prims.begin());
                done = mid > first && mid < first + count;
            }
        }
        if (!done) {
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [axis](const Prim& a, const Prim& b) { return a.cen[axis] < b.cen[axis]; });
        }
        uint32_t left = (uint32_t)nodes.size();
        nodes.push_back(BvhNode{});
        nodes.push_back(BvhNode{});
        nodes[node].leftFirst = left;
        nodes[node].count = 0;
        build(left, first, mid - first, depth + 1);
        build(left + 1, mid, first + count - mid, depth + 1);
    }
};

// builds a subtree rooted at a fresh node appended to `nodes`; returns (root, depth)
std::pair<uint32_t, uint32_t> buildTree(std::vector<Prim>& prims, std::vector<BvhNode>& nodes,
                                        uint32_t leafMax, uint32_t depthLimit) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t mark = nodes.size();
        std::vector<Prim> work = prims;
        nodes.push_back(BvhNode{});
        Builder b{work, nodes, leafMax, attempt == 1};
        b.build((uint32_t)mark, 0, (uint32_t)work.size(), 0);
        if (b.maxDepth <= depthLimit || attempt == 1) {
            prims.swap(work);
            return {(uint32_t)mark, b.maxDepth};
        }
        nodes.resize(mark); // too deep for the LDS stack: rebuild balanced (median splits)
    }
    return {0, 0};
}

} // namespace

void invertAffine(const float m[12], float out[12]) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    double r[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                   (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                   (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    double tx = m[3], ty = m[7], tz = m[11];
    for (int k = 0; k < 3; ++k) {
        out[k * 4 + 0] = (float)r[k * 3 + 0];
        out[k * 4 + 1] = (float)r[k * 3 + 1];
        out[k * 4 + 2] = (float)r[k * 3 + 2];
        out[k * 4 + 3] = (float)(-(r[k * 3 + 0] * tx + r[k * 3 + 1] * ty + r[k * 3 + 2] * tz));
    }
}

// ---- quantised pair records (thx_layout.h: NodePair, TreeFrame) ------------------------------

namespace {
// coordinate a 16-bit value stands for; the same expression the conservative rounding below checks
inline float dequant(uint32_t q, float step, float origin) { return std::fmaf((float)q, step, origin); }

// returns the stack entries a walk of the tree needs (4-wide records; 0 for pair records: the depth of the tree, known to the caller)
uint32_t emitTree(BuiltScene& out, uint32_t root, TreeFrame& frame, bool tlas) {
    const BvhNode& rn = out.nodes[root];
    for (int k = 0; k < 3; ++k) {
        const float lo = rn.lo[k], hi = rn.hi[k];
        // 65531 steps span the extent, from q = 2 to q = 65533: the "one more step outwards" of every plane stays inside 16 bit
        // AND clear of the clamps at both ends of the range (the walk's decoding is good to half a step: dev_trace.hpp, QRay)
        float step = (hi - lo) / 65531.0f;
        step = std::max(step, 1e-7f * std::max(std::fabs(lo), std::fabs(hi)) + 1e-30f);
        float origin = lo - 2.0f * step;
        while (dequant(2u, step, origin) > lo) origin = std::nextafter(origin, -INFINITY);
        while (dequant(65533u, step, origin) < hi) step = std::nextafter(step, INFINITY);
        frame.origin[k] = origin;
        frame.step[k] = step;
    }
    frame.pad = 0;
    auto qlo = [&](float v, int k) { // largest q with dequant(q) <= v, one more step outwards
        float f = std::floor((v - frame.origin[k]) / frame.step[k]);
        int64_t q = (int64_t)std::min(std::max(f, 0.0f), 65534.0f);
        while (q > 0 && dequant((uint32_t)q, frame.step[k], frame.origin[k]) > v) --q;
        return (uint32_t)std::max<int64_t>(q - 1, 0);
    };
    auto qhi = [&](float v, int k) { // smallest q with dequant(q) >= v, one more step outwards
        float f = std::ceil((v - frame.origin[k]) / frame.step[k]);
        int64_t q = (int64_t)std::min(std::max(f, 0.0f), 65534.0f);
        while (q < 65534 && dequant((uint32_t)q, frame.step[k], frame.origin[k]) < v) ++q;
        return (uint32_t)std::min<int64_t>(q + 1, 65535);
    };
    if (out.wide) {
        // 4-wide records (two NodePair slots = one 64-B line each): an inner node's children, the larger inner ones of them replaced
        // by their own children until there are four. An empty slot is a box no ray can hit (lo > hi). Returns what a walk of this
        // tree needs of the stack: a visit pushes up to (children - 1) entries before it descends.
        auto area = [&](uint32_t node) {
            const BvhNode& n = out.nodes[node];
            const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        auto refOf4 = [&](uint32_t node, uint32_t& need, auto&& self) -> uint32_t {
            const BvhNode& n = out.nodes[node];
            need = 0;
            if (n.count != 0) return ((tlas ? out.tlasInstance[n.leftFirst] : n.leftFirst) << 3) | n.count;
            uint32_t ch[4] = {n.leftFirst, n.leftFirst + 1, 0, 0};
            int k = 2;
            while (k < 4) {
                int open = -1;
                float best = -1.0f;
                for (int c = 0; c < k; ++c)
                    if (out.nodes[ch[c]].count == 0 && area(ch[c]) > best) { best = area(ch[c]); open = c; }
                if (open < 0) break;
                const uint32_t left = out.nodes[ch[open]].leftFirst;
                for (int c = k; c > open + 1; --c) ch[c] = ch[c - 1];
                ch[open] = left;
                ch[open + 1] = left + 1;
                ++k;
            }
            const uint32_t index = (uint32_t)out.pairs.size() / 2;
            out.pairs.emplace_back();
            out.pairs.emplace_back();
            uint32_t w[16];
            for (int c = 0; c < 4; ++c) {
                if (c >= k) {
                    w[4 * c + 0] = w[4 * c + 1] = w[4 * c + 2] = 0x0000FFFFu; // lo = 65535, hi = 0
                    w[4 * c + 3] = 0xFFFFFFFFu;
                    continue;
                }
                const BvhNode& cn = out.nodes[ch[c]];
                uint32_t l[3], h[3];
                for (int a = 0; a < 3; ++a) { l[a] = qlo(cn.lo[a], a); h[a] = qhi(cn.hi[a], a); }
                for (int a = 0; a < 3; ++a) w[4 * c + a] = l[a] | (h[a] << 16);
            }
            uint32_t deepest = 0;
            for (int c = 0; c < k; ++c) {
                uint32_t sub = 0;
                w[4 * c + 3] = self(ch[c], sub, self);
                deepest = std::max(deepest, sub);
            }
            std::memcpy(out.pairs[2 * (size_t)index].w, w, 32);
            std::memcpy(out.pairs[2 * (size_t)index + 1].w, w + 8, 32);
            need = (uint32_t)(k - 1) + deepest;
            return index << 3;
        };
        uint32_t need = 0;
        frame.rootRef = refOf4(root, need, refOf4);
        return need;
    }
    // inner nodes of this tree get consecutive pair indices in depth-first order
    auto refOf = [&](uint32_t node, auto&& self) -> uint32_t {
        const BvhNode& n = out.nodes[node];
        if (n.count != 0) return ((tlas ? out.tlasInstance[n.leftFirst] : n.leftFirst) << 3) | n.count; // leaf: as the walk's `cur`
        const uint32_t index = (uint32_t)out.pairs.size();
        out.pairs.emplace_back();
        uint32_t w[8];
        for (int c = 0; c < 2; ++c) {
            const BvhNode& ch = out.nodes[n.leftFirst + c];
            uint32_t l[3], h[3];
            for (int k = 0; k < 3; ++k) { l[k] = qlo(ch.lo[k], k); h[k] = qhi(ch.hi[k], k); }
            for (int k = 0; k < 3; ++k) w[4 * c + k] = l[k] | (h[k] << 16);
        }
        w[3] = self(n.leftFirst, self);
        w[7] = self(n.leftFirst + 1, self);
        std::memcpy(out.pairs[index].w, w, sizeof(w));
        return index << 3;
    };
    frame.rootRef = refOf(root, refOf);
    return 0;
}

void emitPairs(BuiltScene& out) {
    out.pairs.clear();
    out.frames.assign(1 + out.blasRoot.size(), TreeFrame{});
    const uint32_t top = emitTree(out, out.tlasRoot, out.frames[0], true);
    uint32_t deepest = 0;
    for (size_t m = 0; m < out.blasRoot.size(); ++m) deepest = std::max(deepest, emitTree(out, out.blasRoot[m], out.frames[1 + m], false));
    if (out.pairs.empty()) out.pairs.resize(2); // single-leaf trees only: keep the array non-empty
    out.walkInstances.resize(out.instances.size());
    for (size_t k = 0; k < out.instances.size(); ++k) {
        WalkInstance& w = out.walkInstances[k];
        const TreeFrame& f = out.frames[out.instances[k].frame];
        std::memcpy(w.w2o, out.instances[k].w2o, sizeof(w.w2o));
        std::memcpy(w.origin, f.origin, sizeof(w.origin));
        std::memcpy(w.step, f.step, sizeof(w.step));
        w.rootRef = f.rootRef;
        w.pad = 0;
    }
    if (out.wide) out.maxDepth = top + 1 + deepest;
}
} // namespace

BuiltScene buildScene(const thx_mesh_desc* meshes, uint32_t n_meshes,
                      const thx_instance_desc* instances, uint32_t n_instances,
                      uint32_t max_stack_depth) {
    if (n_meshes == 0 || n_instances == 0) throw std::runtime_error("scene cannot be empty");
    BuiltScene out;
    // reserve the TLAS at the front: at most 2*n-1 nodes
    const uint32_t tlasMax = 2 * n_instances;
    out.nodes.resize(tlasMax);

    const uint32_t blasLimit = max_stack_depth > 14 ? (max_stack_depth - 2) / 2 + 2 : 8;
    uint32_t deepestBlas = 0;
    for (uint32_t m = 0; m < n_meshes; ++m) {
        const thx_mesh_desc& md = meshes[m];
        if (!md.vertices || !md.indices || md.n_triangles == 0 || md.n_vertices == 0)
            throw std::runtime_error("mesh without vertices/triangles");
        std::vector<Prim> prims(md.n_triangles);
        for (uint32_t t = 0; t < md.n_triangles; ++t) {
            Prim& p = prims[t];
            p.id = t;
            p.box.reset();
            for (int v = 0; v < 3; ++v) {
                uint32_t vi = md.indices[3 * (size_t)t + v];
                if (vi >= md.n_vertices) throw std::runtime_error("triangle index out of range");
                p.box.grow(md.vertices + 6 * (size_t)vi);
            }
            for (int k = 0; k < 3; ++k) p.cen[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
        }
        std::vector<BvhNode> local;
        // The walk handles up to 4 triangles per leaf (two pairs), but a wavefront pays for the fullest leaf any of its
        // lanes holds: small leaves win although they deepen the tree -- 2 up to a million triangles (round 3, with the cheaper
        // inner-node visit: 81 920 triangles -2.3 % against 3), 3 beyond, where 2 is no faster and the hierarchy a third larger
        // (profiles/r02_park.txt, profiles/r03_inner_loop.txt item 19).
        uint32_t leafMax = md.n_triangles > kSmallMeshTriangles ? 3u : 2u;
        if (const char* e = std::getenv("THX_LEAF_MAX")) leafMax = std::min(4u, std::max(1u, (uint32_t)std::atoi(e)));
        auto [root, depth] = buildTree(prims, local, leafMax, blasLimit);
        (void)root;
        deepestBlas = std::max(deepestBlas, depth);
        // relocate into the global arrays
        const uint32_t nodeBase = (uint32_t)out.nodes.size();
        const uint32_t triBase = (uint32_t)out.tris.size();
        for (BvhNode n : local) {
            if (n.count == 0) n.leftFirst += nodeBase;
            else n.leftFirst += triBase;
            out.nodes.push_back(n);
        }
        out.blasRoot.push_back(nodeBase);
        for (const Prim& p : prims) {
            TriRecord tr;
            std::memset(&tr, 0, sizeof(tr));
            const uint32_t* ix = md.indices + 3 * (size_t)p.id;
            for (int k = 0; k < 3; ++k) { // x y z x y
                const float* vtx = md.vertices + 6 * (size_t)ix[k];
                std::memcpy(tr.v[k], vtx, 12);
                std::memcpy(tr.v[k] + 3, vtx, 8);
            }
            tr.prim = p.id;
            out.tris.push_back(tr);
            NormalRecord nr;
            std::memset(&nr, 0, sizeof(nr));
            std::memcpy(nr.n0, md.vertices + 6 * (size_t)ix[0] + 3, 12);
            std::memcpy(nr.n1, md.vertices + 6 * (size_t)ix[1] + 3, 12);
            std::memcpy(nr.n2, md.vertices + 6 * (size_t)ix[2] + 3, 12);
            out.normals.push_back(nr);
        }
    }

    // instances + TLAS
    std::vector<Prim> iprims(n_instances);
    out.instances.resize(n_instances);
    for (uint32_t k = 0; k < n_instances; ++k) {
        const thx_instance_desc& id = instances[k];
        if (id.mesh >= n_meshes) throw std::runtime_error("instance references unknown mesh");
        InstanceRecord& rec = out.instances[k];
        std::memset(&rec, 0, sizeof(rec));
        std::memcpy(rec.o2w, id.transform, sizeof(float) * 12);
        invertAffine(rec.o2w, rec.w2o);
        for (int q = 0; q < 12; ++q)
            if (!std::isfinite(rec.w2o[q])) throw std::runtime_error("singular instance transform");
        rec.blasRoot = out.blasRoot[id.mesh];
        rec.customIndex = id.customIndex;
        rec.geometry = k;
        Prim& p = iprims[k];
        p.id = k;
        p.box.reset();
        const thx_mesh_desc& md = meshes[id.mesh];
        for (uint32_t v = 0; v < md.n_vertices; ++v) {
            const float* q = md.vertices + 6 * (size_t)v;
            float w[3];
            for (int a = 0; a < 3; ++a)
                w[a] = rec.o2w[a * 4 + 3] + (rec.o2w[a * 4 + 0] * q[0] + rec.o2w[a * 4 + 1] * q[1] + rec.o2w[a * 4 + 2] * q[2]);
            p.box.grow(w);
        }
        for (int a = 0; a < 3; ++a) { // transform rounding slack
            float pad = 4e-6f * std::max(std::fabs(p.box.lo[a]), std::fabs(p.box.hi[a])) + 1e-30f;
            p.box.lo[a] -= pad;
            p.box.hi[a] += pad;
            p.cen[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        }
    }
    std::vector<BvhNode> tl;
    uint32_t tlasLimit = max_stack_depth - deepestBlas - 2;
    auto [troot, tdepth] = buildTree(iprims, tl, 1, tlasLimit);
    (void)troot;
    if (tl.size() > tlasMax) throw std::runtime_error("internal: TLAS larger than reserved");
    for (size_t i = 0; i < tl.size(); ++i) out.nodes[i] = tl[i]; // child indices are already absolute (base 0)
    // unused reserved slots stay zero-filled and unreachable
    out.tlasRoot = 0;
    out.tlasInstance.resize(n_instances);
    for (uint32_t s = 0; s < n_instances; ++s) out.tlasInstance[s] = iprims[s].id;
    out.maxDepth = tdepth + 1 + deepestBlas;
    if (out.maxDepth + 1 > max_stack_depth)
        throw std::runtime_error("scene hierarchy deeper than the traversal stack");
    for (uint32_t k = 0; k < n_instances; ++k) out.instances[k].frame = 1 + instances[k].mesh;
    emitPairs(out);
    // Large scenes get 4-wide records (bvh_build.h: kWideFromPairs; THX_WIDE=0 / 1: never / always -- the walk's forms are tested on small scenes)
    bool wantWide = out.pairs.size() >= kWideFromPairs;
    if (const char* e = std::getenv("THX_WIDE")) wantWide = std::atoi(e) != 0;
    if (wantWide) {
        const uint32_t binaryDepth = out.maxDepth;
        out.wide = true;
        emitPairs(out);
        if (out.maxDepth + 1 > max_stack_depth) { // (three entries per 4-wide level: deeper than the stack bound -> stay binary)
            out.wide = false;
            out.maxDepth = binaryDepth;
            emitPairs(out);
        }
    }
    return out;
}

} // namespace thx
