// Local map of tracked instances and relocalization: frontend/GraphMap.{hpp,cpp},
// core/CommonUtils.cpp:115-204,264-316, pipeline/SemGraphSLAM.cpp:54,81-107,116.  Tens to a few hundred nodes per
// scan: host code.  The graphs themselves (BuildGraph / ReBuildGraph) come from the device-assisted builder the
// hot path already has; it is handed in as a callback so this header stays free of the context type.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <random>
#include <unordered_map>
#include <vector>

#include "sgs_math.h"

namespace sgs {

// what GraphMap needs of a Graph (core/Coreutils.h:59-78)
struct GraphData {
    int n = 0, D = 0;
    std::vector<double> boxes;        // n x 8: centre, dimension, label, points (sgs_cluster_points layout)
    std::vector<float> desc;          // n x D node descriptors
    std::vector<int32_t> tri_counts;  // sub-triangles per node
    std::vector<double> tri;          // all sub-triangles, 3 sorted edge lengths each, node after node
    std::vector<size_t> tri_off;      // first triangle of each node
    int label(int i) const { return (int)boxes[8 * (size_t)i + 6]; }
    int points(int i) const { return (int)boxes[8 * (size_t)i + 7]; }
    D3 centre(int i) const { return D3{boxes[8 * (size_t)i], boxes[8 * (size_t)i + 1], boxes[8 * (size_t)i + 2]}; }
    D3 dim(int i) const { return D3{boxes[8 * (size_t)i + 3], boxes[8 * (size_t)i + 4], boxes[8 * (size_t)i + 5]}; }
    void index_triangles() {
        tri_off.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) tri_off[(size_t)i + 1] = tri_off[(size_t)i] + (size_t)tri_counts[(size_t)i];
    }
};
using GraphBuilder = std::function<int(const double *boxes, int nb, GraphData *out)>;

struct MapInstance {  // InsNode, core/Coreutils.h:81-87
    int label, id;
    D3 pose, dim;
    int points;
};
struct InsMatch {  // an entry of VTbii, GraphMap.hpp:23
    int flag, frame_id, map_id;
};

// GraphMap.cpp:11-40
inline bool check_sub_triangle(const GraphData &a, int i, const GraphData &b, int j) {
    const size_t na = (size_t)a.tri_counts[(size_t)i], nb = (size_t)b.tri_counts[(size_t)j];
    if (na < 2 && nb < 2) return true;
    int matched = 0;
    for (size_t m = 0; m < na; ++m) {
        const double *p = &a.tri[3 * (a.tri_off[(size_t)i] + m)];
        for (size_t q = 0; q < nb; ++q) {
            const double *r = &b.tri[3 * (b.tri_off[(size_t)j] + q)];
            if (std::fabs(p[0] - r[0]) < 0.3 && std::fabs(p[1] - r[1]) < 0.3 && std::fabs(p[2] - r[2]) < 0.3) matched++;
        }
    }
    if (na > 5 && nb > 5) return matched >= 3;
    return matched >= 1;
}

// float squared distance, four terms at a time (nanoflann's L2 adaptor, core/nanoflann.hpp)
inline float desc_dist2(const float *a, const float *b, int n) {
    float result = 0.f;
    int d = 0;
    for (; d + 3 < n; d += 4) {
        const float d0 = a[d] - b[d], d1 = a[d + 1] - b[d + 1], d2 = a[d + 2] - b[d + 2], d3 = a[d + 3] - b[d + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; d < n; ++d) {
        const float d0 = a[d] - b[d];
        result += d0 * d0;
    }
    return result;
}

// GraphMap.cpp:88-113 FindInsMatch (with :43-82 FindCorrespondencesKDtree: exact 5-NN in descriptor space, so a
// brute-force scan in index order gives the kd-tree's neighbours; first one with equal label and compatible
// sub-triangles wins; result slots the search could not fill stay at index 0 like the reference's zeroed vector)
inline std::vector<InsMatch> find_ins_match(const GraphData &frame, const GraphData &map) {
    std::vector<InsMatch> out;
    if (map.n == 0) {
        for (int i = 0; i < frame.n; ++i) out.push_back({0, i, -1});
        return out;
    }
    std::vector<char> is_match((size_t)frame.n, 0);
    std::vector<std::pair<float, int>> d((size_t)map.n);
    for (int i = 0; i < frame.n; ++i) {
        for (int j = 0; j < map.n; ++j)
            d[(size_t)j] = {desc_dist2(&frame.desc[(size_t)i * frame.D], &map.desc[(size_t)j * map.D], frame.D), j};
        const int k = std::min(5, map.n);
        std::partial_sort(d.begin(), d.begin() + k, d.end(), [](const std::pair<float, int> &a, const std::pair<float, int> &b) {
            return a.first < b.first || (a.first == b.first && a.second < b.second);
        });
        for (int m = 0; m < 5; ++m) {
            const int j = m < map.n ? d[(size_t)m].second : 0;
            if (frame.label(i) == map.label(j) && check_sub_triangle(frame, i, map, j)) {
                out.push_back({1, i, j});
                is_match[(size_t)i] = 1;
                break;
            }
        }
    }
    for (int i = 0; i < frame.n; ++i)
        if (!is_match[(size_t)i]) out.push_back({0, i, -1});
    return out;
}

struct M3 {
    double m[3][3];
};
inline D3 mul(const M3 &R, D3 p) {
    return D3{R.m[0][0] * p.x + R.m[0][1] * p.y + R.m[0][2] * p.z, R.m[1][0] * p.x + R.m[1][1] * p.y + R.m[1][2] * p.z,
              R.m[2][0] * p.x + R.m[2][1] * p.y + R.m[2][2] * p.z};
}
struct Iso {  // Eigen::Isometry3d
    M3 R;
    D3 t;
};
inline M3 m3_identity() { return M3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }

// CommonUtils.cpp:264-316 SolveSVD: R = V E U^T from W = sum (p1 - mean1)(p2 - mean2)^T.  The SVD is a one-sided
// Jacobi iteration on W itself; directions W does not span (three centred points are coplanar) are completed by
// cross products, the determinant rule then fixes the sign.  Less than two independent directions: R = I.
inline Iso solve_svd(const D3 *n1, const D3 *n2, int N) {
    D3 s1{0, 0, 0}, s2{0, 0, 0};
    for (int i = 0; i < N; ++i) {
        s1 = add(s1, n1[i]);
        s2 = add(s2, n2[i]);
    }
    const D3 m1{s1.x / N, s1.y / N, s1.z / N}, m2{s2.x / N, s2.y / N, s2.z / N};
    double W[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < N; ++i) {
        const D3 a = sub(n1[i], m1), b = sub(n2[i], m2);
        const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) W[r][c] += av[r] * bv[c];
    }
    // one-sided (Hestenes) Jacobi SVD of W: rotate column pairs of A = W V until they are orthogonal; then the
    // column norms are the singular values, the normalised columns U, and V the accumulated rotations
    double A[3][3], V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            A[r][c] = W[r][c];
            V[r][c] = r == c ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += A[k][p] * A[k][p];
                    beta += A[k][q] * A[k][q];
                    gamma += A[k][p] * A[k][q];
                }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-16 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
    double sv[3];
    for (int j = 0; j < 3; ++j) sv[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return sv[a] > sv[b]; });
    D3 v[3], u[3];
    for (int k = 0; k < 3; ++k) v[k] = D3{V[0][ord[k]], V[1][ord[k]], V[2][ord[k]]};
    const double smax = sv[ord[0]];
    int rank = 0;
    for (int k = 0; k < 3; ++k) {
        const double sk = sv[ord[k]];
        if (!(sk > 1e-9 * smax && sk > 1e-150)) break;
        u[k] = D3{A[0][ord[k]] / sk, A[1][ord[k]] / sk, A[2][ord[k]] / sk};
        rank = k + 1;
    }
    Iso T;
    T.R = m3_identity();
    if (rank >= 2) {
        if (rank == 2) {
            u[2] = crs(u[0], u[1]);
            v[2] = crs(v[0], v[1]);
        }
        const double vv[3][3] = {{v[0].x, v[1].x, v[2].x}, {v[0].y, v[1].y, v[2].y}, {v[0].z, v[1].z, v[2].z}};
        const double uu[3][3] = {{u[0].x, u[1].x, u[2].x}, {u[0].y, u[1].y, u[2].y}, {u[0].z, u[1].z, u[2].z}};
        auto build = [&](double e) {
            M3 R;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R.m[r][c] = vv[r][0] * uu[c][0] + vv[r][1] * uu[c][1] + e * vv[r][2] * uu[c][2];
            return R;
        };
        M3 R = build(1.0);
        const double det = R.m[0][0] * (R.m[1][1] * R.m[2][2] - R.m[1][2] * R.m[2][1]) - R.m[0][1] * (R.m[1][0] * R.m[2][2] - R.m[1][2] * R.m[2][0]) +
                           R.m[0][2] * (R.m[1][0] * R.m[2][1] - R.m[1][1] * R.m[2][0]);
        if (det < 0) R = build(-1.0);
        T.R = R;
    }
    T.t = sub(m2, mul(T.R, m1));
    return T;
}

// Eigen's Quaternion(Matrix3) + normalisation: what Sophus::SE3d(Matrix4d) keeps of the rotation (GraphMap.cpp:140)
inline Rigid rigid_from_iso(const Iso &T) {
    const M3 &R = T.R;
    double w, x, y, z;
    const double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        w = 0.5 * s;
        s = 0.5 / s;
        x = (R.m[2][1] - R.m[1][2]) * s; y = (R.m[0][2] - R.m[2][0]) * s; z = (R.m[1][0] - R.m[0][1]) * s;
    } else {
        int i = 0;
        if (R.m[1][1] > R.m[0][0]) i = 1;
        if (R.m[2][2] > R.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
        double v[3];
        v[i] = 0.5 * s;
        s = 0.5 / s;
        w = (R.m[k][j] - R.m[j][k]) * s;
        v[j] = (R.m[j][i] + R.m[i][j]) * s;
        v[k] = (R.m[k][i] + R.m[i][k]) * s;
        x = v[0]; y = v[1]; z = v[2];
    }
    const double n = std::sqrt(w * w + x * x + y * y + z * z);
    return Rigid{w / n, x / n, y / n, z / n, T.t};
}

struct KeyHash {
    size_t operator()(const K3 &k) const { return (size_t)ref_hash(k); }
};
struct KeyEq {
    bool operator()(const K3 &a, const K3 &b) const { return a.x == b.x && a.y == b.y && a.z == b.z; }
};

// CommonUtils.cpp:115-204 RansacAlignment(match1, match2, graph1, graph2, max_iter, best): 500 random triplets; a
// frame node is an inlier when one of the 27 voxels (0.2 m, held in a float) around its image holds a map node.
// The reference seeds std::mt19937 from std::random_device; the seed is the caller's here.
inline Iso ransac_alignment(const std::vector<D3> &m1, const std::vector<D3> &m2, const std::vector<D3> &nodes1, const std::vector<D3> &nodes2,
                            int max_iter, uint32_t seed, int *best_inlier, double *score) {
    std::mt19937 gen(seed);
    std::uniform_int_distribution<> dis(0, (int)m1.size() - 1);
    Iso final_T{m3_identity(), D3{0, 0, 0}};
    const float voxel_size = 0.2f;
    std::unordered_map<K3, int, KeyHash, KeyEq> occupied;
    for (const D3 &p : nodes2) occupied.emplace(K3{(int)(p.x / voxel_size), (int)(p.y / voxel_size), (int)(p.z / voxel_size)}, 1);
    for (int it = 0; it < max_iter; ++it) {
        const int i1 = dis(gen), i2 = dis(gen), i3 = dis(gen);
        const D3 a[3] = {m1[(size_t)i1], m1[(size_t)i2], m1[(size_t)i3]}, b[3] = {m2[(size_t)i1], m2[(size_t)i2], m2[(size_t)i3]};
        const Iso T = solve_svd(a, b, 3);
        int inliers = 0;
        for (const D3 &p : nodes1) {
            const D3 q = add(mul(T.R, p), T.t);
            const int kx = (int)(q.x / voxel_size), ky = (int)(q.y / voxel_size), kz = (int)(q.z / voxel_size);
            bool hit = false;
            for (int i = kx - 1; i <= kx + 1 && !hit; ++i)
                for (int j = ky - 1; j <= ky + 1 && !hit; ++j)
                    for (int k = kz - 1; k <= kz + 1 && !hit; ++k) hit = occupied.count(K3{i, j, k}) != 0;
            if (hit) inliers++;
        }
        if (inliers > *best_inlier) {
            *best_inlier = inliers;
            final_T = T;
        }
    }
    *score = (float)*best_inlier / nodes1.size();
    return final_T;
}

struct GraphMapParams {
    double voxel_size = 1.5;  // GraphMap.hpp:31
    double nn_vehicle = 1.0, nn_pole = 1.0, max_distance = 100.0;
};

struct GraphMapState {  // GraphMap.hpp:28-71
    std::vector<MapInstance> inst;  // instance_in_localmap
    int next_id = 0;                // instance_id_
    GraphData graph;                // InsGraphMap
    void clear() {
        inst.clear();
        next_id = 0;
        graph = GraphData{};
    }
};

// GraphMap.cpp:149-222 Update.  fresh: frame node ids that became new instances (Graph::new_instance).
inline int graph_map_update(GraphMapState &gm, const GraphMapParams &P, const GraphData &frame, const std::vector<InsMatch> &matches,
                            const Rigid &pose, const GraphBuilder &rebuild, std::vector<int> *fresh) {
    // voxel -> instances, built once from the instances present before this update (as the reference)
    std::unordered_map<K3, std::vector<int>, KeyHash, KeyEq> hash;
    auto key = [&](D3 p) { return K3{(int)(p.x / P.voxel_size), (int)(p.y / P.voxel_size), (int)(p.z / P.voxel_size)}; };
    for (int idx = 0; idx < (int)gm.inst.size(); ++idx) hash[key(gm.inst[(size_t)idx].pose)].push_back(idx);
    auto near_one = [&](D3 p, int label, double th) {  // :227-262 IsExistNearestNeighbor
        const K3 c = key(p);
        for (int i = c.x - 1; i <= c.x + 1; ++i)
            for (int j = c.y - 1; j <= c.y + 1; ++j)
                for (int k = c.z - 1; k <= c.z + 1; ++k) {
                    const auto f = hash.find(K3{i, j, k});
                    if (f == hash.end()) continue;
                    for (int idx : f->second) {
                        if (idx < 0 || idx >= (int)gm.inst.size()) continue;
                        if (gm.inst[(size_t)idx].label != label) continue;
                        if (std::sqrt(sq3(sub(p, gm.inst[(size_t)idx].pose))) < th) return true;
                    }
                }
        return false;
    };
    for (const InsMatch &m : matches) {
        bool matched = m.flag != 0;
        const int label = frame.label(m.frame_id);
        const double th = label == 1 ? P.nn_vehicle : P.nn_pole;
        if (matched) {
            const D3 p = apply(pose, frame.centre(m.frame_id));
            MapInstance &ins = gm.inst[(size_t)m.map_id];
            if (std::sqrt(sq3(sub(p, ins.pose))) < th) {
                ins.pose = D3{(ins.pose.x + p.x) / 2, (ins.pose.y + p.y) / 2, (ins.pose.z + p.z) / 2};
                const D3 d = frame.dim(m.frame_id);
                ins.dim = D3{(ins.dim.x + d.x) / 2, (ins.dim.y + d.y) / 2, (ins.dim.z + d.z) / 2};
            } else {
                matched = false;
            }
        }
        if (!matched) {
            const D3 d = frame.dim(m.frame_id);
            if (label == 1) {
                if (frame.points(m.frame_id) < 50) continue;
                if (d.x < 1 || d.y < 1 || d.z < 1) continue;
            } else if (label == 2 || label == 3) {
                if (d.z < 1) continue;
            }
            MapInstance ins{label, gm.next_id++, apply(pose, frame.centre(m.frame_id)), d, frame.points(m.frame_id)};
            if (near_one(ins.pose, ins.label, P.nn_pole)) continue;
            if (fresh) fresh->push_back(m.frame_id);
            gm.inst.push_back(ins);
        }
    }
    const double max_d2 = P.max_distance * P.max_distance;
    std::vector<MapInstance> kept;
    for (const MapInstance &ins : gm.inst)
        if (!(sq3(sub(ins.pose, pose.t)) > max_d2)) kept.push_back(ins);
    gm.inst.swap(kept);
    // ReBuildGraph (SemGraph.cpp:113-208) = BuildGraph on the instances
    std::vector<double> boxes(gm.inst.size() * 8);
    for (size_t i = 0; i < gm.inst.size(); ++i) {
        const MapInstance &v = gm.inst[i];
        double *b = &boxes[8 * i];
        b[0] = v.pose.x; b[1] = v.pose.y; b[2] = v.pose.z; b[3] = v.dim.x; b[4] = v.dim.y; b[5] = v.dim.z; b[6] = v.label; b[7] = v.points;
    }
    return rebuild(boxes.data(), (int)gm.inst.size(), &gm.graph);
}

// GraphMap.cpp:118-144 Relocalization
inline bool graph_map_relocalize(const GraphMapState &gm, const GraphData &frame, const std::vector<InsMatch> &matches, double inlier_rate_th,
                                 uint32_t seed, Rigid *out, double *score_out, int *n_matched) {
    std::vector<D3> m1, m2, nodes1, nodes2;
    for (const InsMatch &m : matches)
        if (m.flag) {
            m1.push_back(frame.centre(m.frame_id));
            m2.push_back(gm.graph.centre(m.map_id));
        }
    if (n_matched) *n_matched = (int)m1.size();
    *out = rigid_identity();
    double score = 0;
    bool ok = false;
    if (m1.size() > 3) {
        for (int i = 0; i < frame.n; ++i) nodes1.push_back(frame.centre(i));
        for (int i = 0; i < gm.graph.n; ++i) nodes2.push_back(gm.graph.centre(i));
        int best = 0;
        const Iso T = ransac_alignment(m1, m2, nodes1, nodes2, 500, seed, &best, &score);
        ok = score > inlier_rate_th;
        *out = rigid_from_iso(T);
    }
    if (score_out) *score_out = score;
    return ok;
}

}  // namespace sgs
