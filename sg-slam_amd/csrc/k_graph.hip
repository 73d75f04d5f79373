// Semantic graph on device.
//
// ClusterPoints (F/SemanticCluster.cpp:205-360): route the 0.2 m cloud to the five foreground classes
// {1,4,5,16,18} (order-preserving multi-way compaction), then per class the curved-voxel clustering
// of SemanticCluster (:22-174).  The reference's cluster() is an ORDER-DEPENDENT greedy labelling
// (labelled points never expand, :113-114), not connected components, so it is executed exactly:
// one workgroup per class builds the polar/range/azimuth cell table in parallel (integer cell keys,
// open-addressing table, per-cell point lists sorted by input index), then ONE WAVEFRONT walks the
// seeds in input order; for each seed the 27 neighbour cells are looked up by lanes 0..26 and the
// neighbour points are examined 64 at a time with ballots.  "relabel every oc -> nc"
// (mergeClusters, :101-107) becomes parent[oc] = nc in a union-find over label VALUES, which yields
// the same final label values as the reference's O(n) relabel scans.
// Per-cluster count / AABB / first index are reduced with integer atomics on order-preserving
// encodings of the doubles; the host then replays std::unordered_map<int,...> insertion to emit the
// boxes in the reference's (libstdc++) iteration order (:305-307).
//
// BuildGraph (F/SemGraph.cpp:14-62): pairwise centre distances, adjacency and the per-node
// 3 x subinterval edge-length/class histogram in one kernel (LDS histogram per node row); the
// spectral half (MatrixDecomposing, :215-240) and the sub-triangles are O(N^3)/tiny and stay on host.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <numeric>
#include <unordered_map>

#include "sgs_graph.h"

namespace sgs {

namespace {

constexpr int kBlock = 256;

__device__ __host__ inline int fg_class_of(int label) {
    switch (label) {
        case 1: return 0;
        case 4: return 1;
        case 5: return 2;
        case 16: return 3;
        case 18: return 4;
        default: return -1;
    }
}
__device__ inline bool is_background(int label) { return label == 14 || label == 15 || label == 9 || label == 13; }

// ---- routing: 7 categories = 5 classes, any-foreground, background ----
__global__ __launch_bounds__(kBlock) void route_count_kernel(CloudView in, const int32_t *__restrict__ n_ptr,
                                                             int32_t *__restrict__ blockcnt, int32_t *__restrict__ pt_cluster) {
    const int n = *n_ptr;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int cls = -1;
    bool bg = false;
    if (i < n) {
        const int L = in.l[i];
        cls = fg_class_of(L);
        bg = is_background(L);
        pt_cluster[i] = 0;
    }
    __shared__ int cnt[8];
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const bool f = k < 5 ? (cls == k) : (k == 5 ? cls >= 0 : bg);
        const unsigned long long b = __ballot(f);
        if (lane == 0 && b) atomicAdd(&cnt[k], __popcll(b));
    }
    __syncthreads();
    if (threadIdx.x < 8) blockcnt[blockIdx.x * 8 + threadIdx.x] = cnt[threadIdx.x];
}

// exclusive per-category prefix over blocks (in place) + totals + class parameters
__global__ __launch_bounds__(kBlock) void route_totals_kernel(int32_t *__restrict__ blockcnt, int nblocks,
                                                              int32_t *__restrict__ counts, ClassParams *__restrict__ params) {
    __shared__ int tot[8];
    __shared__ int part[kBlock];
    const int per = (nblocks + kBlock - 1) / kBlock;
    const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
    for (int k = 0; k < 7; ++k) {
        int sum = 0;
        for (int b = b0; b < b1; ++b) sum += blockcnt[b * 8 + k];
        part[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int t = 0; t < kBlock; ++t) {
                const int v = part[t];
                part[t] = run;
                run += v;
            }
            tot[k] = run;
            counts[k] = run;
        }
        __syncthreads();
        int run = part[threadIdx.x];
        for (int b = b0; b < b1; ++b) {
            const int v = blockcnt[b * 8 + k];
            blockcnt[b * 8 + k] = run;
            run += v;
        }
        __syncthreads();
    }
    const int k = threadIdx.x;
    if (k == 0) {
        int start = 0;
        for (int c = 0; c < kNumFgClasses; ++c) {
            ClassParams p;
            p.n = tot[c];
            p.start = start;
            p.length = p.width = p.height = 0;
            int cap = 1;
            while (cap < 2 * p.n) cap <<= 1;
            p.table_off = 4 * start;
            p.table_mask = cap - 1;
            p.n_labels = 0;
            p.min_range = 0;
            p.max_range = 0;
            params[c] = p;
            start += p.n;
        }
    }
}

__global__ __launch_bounds__(kBlock) void route_scatter_kernel(CloudView in, const int32_t *__restrict__ n_ptr,
                                                               const int32_t *__restrict__ blockbase,
                                                               const ClassParams *__restrict__ params,
                                                               int32_t *__restrict__ fg_order, int32_t *__restrict__ fg_all,
                                                               int32_t *__restrict__ bg_all) {
    const int n = *n_ptr;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int cls = -1;
    bool bg = false;
    if (i < n) {
        const int L = in.l[i];
        cls = fg_class_of(L);
        bg = is_background(L);
    }
    __shared__ int wcnt[7][kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int myrank[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const bool f = k < 5 ? (cls == k) : (k == 5 ? cls >= 0 : bg);
        const unsigned long long b = __ballot(f);
        myrank[k] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[k][wave] = __popcll(b);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int base = blockbase[blockIdx.x * 8 + k];
        for (int w = 0; w < wave; ++w) base += wcnt[k][w];
        const bool f = k < 5 ? (cls == k) : (k == 5 ? cls >= 0 : bg);
        if (f) {
            if (k < 5) fg_order[params[k].start + base + myrank[k]] = i;
            else if (k == 5) fg_all[base + myrank[k]] = i;
            else bg_all[base + myrank[k]] = i;
        }
    }
}

__global__ __launch_bounds__(kBlock) void stat_init_kernel(ClusterStat *__restrict__ stat, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    ClusterStat s;
    s.lo[0] = s.lo[1] = s.lo[2] = ~0ull;
    s.hi[0] = s.hi[1] = s.hi[2] = 0ull;
    s.count = 0;
    s.first = INT_MAX;
    stat[i] = s;
}

struct ClusterConsts {
    double dA, dR, dP;
    double min_az;  // -24.8 deg in rad, computed on the host exactly as the reference writes it
    int width, height;
};

__device__ inline double polar_angle(double x, double y) {  // F/SemanticCluster.cpp:10-20
    double t = 0;
    if (x == 0 && y == 0) t = 0;
    else if (y >= 0) t = atan2(y, x);
    else if (y < 0) t = atan2(y, x) + 2 * M_PI;
    return t;
}

__device__ inline int find_root(const int32_t *parent, int v) {
    while (true) {
        const int p = parent[v];
        if (p == v) return v;
        v = p;
    }
}

__device__ inline uint32_t cell_hash(int key) { return (uint32_t)key * 2654435761u; }

// one workgroup per class
__global__ __launch_bounds__(kBlock) void cluster_class_kernel(CloudView in, ClassParams *__restrict__ params,
                                                               const int32_t *__restrict__ fg_order, ClusterConsts K,
                                                               int32_t *__restrict__ cell3, int32_t *__restrict__ cellslot,
                                                               int32_t *__restrict__ cell_pts, CellSlot *__restrict__ table_all,
                                                               int32_t *label_all, int32_t *parent_all) {
    const int c = blockIdx.x;
    const int n = params[c].n;
    if (n <= 0) return;
    const int start = params[c].start;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *order = fg_order + start;
    int32_t *label = label_all + start;
    int32_t *parent = parent_all + start + c;  // indexed by label value 1..n
    int32_t *cpts = cell_pts + start;
    int32_t *c3 = cell3 + 4 * (size_t)start;
    int32_t *cslot = cellslot + start;
    CellSlot *table = table_all + params[c].table_off;
    const int tmask = params[c].table_mask;

    __shared__ double s_min[kBlock / 64], s_max[kBlock / 64];
    __shared__ double s_minR, s_maxR;
    __shared__ int s_len;
    __shared__ int s_scan[kBlock];
    __shared__ int s_excl[28], s_cstart[27];

    // ---- A: calculateAPR range extrema (:22-36)
    double mn = 1.7976931348623157e308, mx = 2.2250738585072014e-308;
    for (int j = tid; j < n; j += kBlock) {
        const int oi = order[j];
        const double x = in.x[oi], y = in.y[oi];
        const double r = sqrt(x * x + y * y);
        if (r < mn) mn = r;
        if (r > mx) mx = r;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_down(mn, o), b = __shfl_down(mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; }
    __syncthreads();
    if (tid == 0) {
        double a = s_min[0], b = s_max[0];
        for (int w = 1; w < kBlock / 64; ++w) {
            a = s_min[w] < a ? s_min[w] : a;
            b = s_max[w] > b ? s_max[w] : b;
        }
        s_minR = a; s_maxR = b;
        s_len = (int)((b - a) / K.dR) + 1;  // :37
        params[c].min_range = a; params[c].max_range = b;
        params[c].length = s_len; params[c].width = K.width; params[c].height = K.height;
    }
    for (int s = tid; s <= tmask; s += kBlock) table[s] = CellSlot{INT_MIN, 0, 0, 0};
    __syncthreads();
    const double minR = s_minR;
    const int length = s_len, width = K.width, height = K.height;

    // ---- B: cell of every point (:50-54) + cell table insert
    for (int j = tid; j < n; j += kBlock) {
        const int oi = order[j];
        const double x = in.x[oi], y = in.y[oi], z = in.z[oi];
        const double polar = polar_angle(x, y);
        const double range = sqrt(x * x + y * y);
        const double azimuth = atan2(z, range);
        const int azi = (int)(((azimuth - K.min_az) * 180 / M_PI) / K.dA);
        const int pol = (int)(polar * 180 / M_PI / K.dP);
        const int rng = (int)((range - minR) / K.dR);
        const int vidx = (pol * length + rng) + azi * length * width;
        c3[4 * j] = pol; c3[4 * j + 1] = rng; c3[4 * j + 2] = azi; c3[4 * j + 3] = vidx;
        uint32_t s = cell_hash(vidx) & (uint32_t)tmask;
        while (true) {
            const int cur = atomicCAS(&table[s].key, INT_MIN, vidx);
            if (cur == INT_MIN || cur == vidx) break;
            s = (s + 1u) & (uint32_t)tmask;
        }
        atomicAdd(&table[s].count, 1);
        cslot[j] = (int)s;
        label[j] = -1;
    }
    __syncthreads();

    // ---- C: exclusive scan of the cell counts -> list starts
    const int cap = tmask + 1;
    const int per = (cap + kBlock - 1) / kBlock;
    {
        int sum = 0;
        for (int s = tid * per; s < (tid + 1) * per && s < cap; ++s) sum += table[s].count;
        s_scan[tid] = sum;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int t = 0; t < kBlock; ++t) {
                const int v = s_scan[t];
                s_scan[t] = run;
                run += v;
            }
        }
        __syncthreads();
        int run = s_scan[tid];
        for (int s = tid * per; s < (tid + 1) * per && s < cap; ++s) {
            table[s].start = run;
            run += table[s].count;
        }
    }
    __syncthreads();
    // ---- D: fill the per-cell lists, then sort each by point index (insertion order of :58)
    for (int j = tid; j < n; j += kBlock) {
        const int s = cslot[j];
        const int pos = table[s].start + atomicAdd(&table[s].cursor, 1);
        cpts[pos] = j;
    }
    __syncthreads();
    for (int s = tid; s < cap; s += kBlock) {
        const int cnt = table[s].count, st = table[s].start;
        for (int a = 1; a < cnt; ++a) {
            const int v = cpts[st + a];
            int b = a - 1;
            while (b >= 0 && cpts[st + b] > v) {
                cpts[st + b + 1] = cpts[st + b];
                --b;
            }
            cpts[st + b + 1] = v;
        }
    }
    __syncthreads();
    if (tid >= 64) return;

    // ---- F: the greedy labelling (:109-174), seeds in input order, one wavefront
    int current = 0;
    for (int base = 0; base < n; base += 64) {
        while (true) {
            const int jj = base + lane;
            const int lab = jj < n ? label[jj] : 0;
            const unsigned long long um = __ballot(jj < n && lab == -1);
            if (!um) break;
            const int i = base + __ffsll((long long)um) - 1;  // next unlabelled point = seed
            const int pol = c3[4 * i], rng = c3[4 * i + 1], azi = c3[4 * i + 2];
            // neighbour cells in z / y / x order (find_neighbors :72-99)
            int ccount = 0, cstart = 0;
            if (lane < 27) {
                const int z = azi + lane / 9 - 1, y = rng + (lane / 3) % 3 - 1, x = pol + lane % 3 - 1;
                if (!(z < 0 || z > height - 1) && !(y < 0 || y > length - 1)) {
                    int px = x;
                    if (x < 0) px = width - 1;
                    if (x > width - 1) px = 0;
                    const int key = (px * length + y) + z * length * width;
                    uint32_t s = cell_hash(key) & (uint32_t)tmask;
                    for (int step = 0; step <= tmask; ++step) {
                        const CellSlot e = table[s];
                        if (e.key == key) { ccount = e.count; cstart = e.start; break; }
                        if (e.key == INT_MIN) break;
                        s = (s + 1u) & (uint32_t)tmask;
                    }
                }
            }
            int incl = ccount;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            const int m = __shfl(incl, 26);
            if (lane < 27) { s_excl[lane] = incl - ccount; s_cstart[lane] = cstart; }
            if (lane == 27) s_excl[27] = m;
            __syncthreads();
            int oc = -1;
            for (int t0 = 0; t0 < m; t0 += 64) {
                const int t = t0 + lane;
                const bool valid = t < m;
                int nj = -1, root = -1;
                if (valid) {
                    int k = 0;
#pragma unroll
                    for (int step = 16; step > 0; step >>= 1)
                        if (k + step < 27 && s_excl[k + step] <= t) k += step;
                    nj = cpts[s_cstart[k] + (t - s_excl[k])];
                    const int lj = label[nj];
                    if (lj > 0) root = find_root(parent, lj);
                }
                unsigned long long active = __ballot(valid);
                if (oc == -1) {
                    const unsigned long long labelled = __ballot(valid && root != -1);
                    if (!labelled) continue;  // everything here precedes the first labelled neighbour
                    const int fl = __ffsll((long long)labelled) - 1;
                    oc = __shfl(root, fl);
                    active &= ~((2ull << fl) - 1ull);  // positions after the first labelled one
                }
                const bool mine = (active >> lane) & 1ull;
                while (true) {
                    const unsigned long long diff = __ballot(mine && root != -1 && root != oc);
                    if (!diff) break;
                    const int dl = __ffsll((long long)diff) - 1;
                    const int nc = __shfl(root, dl);
                    if (lane == 0) parent[oc] = nc;  // mergeClusters(oc -> nc)
                    if (root == oc) root = nc;
                    oc = nc;
                }
                if (mine && root == -1) label[nj] = oc;
                __syncthreads();
            }
            if (oc == -1) {  // nobody around was labelled: new cluster for the seed and all neighbours
                ++current;
                if (lane == 0) { parent[current] = current; label[i] = current; }
                for (int t0 = 0; t0 < m; t0 += 64) {
                    const int t = t0 + lane;
                    if (t < m) {
                        int k = 0;
#pragma unroll
                        for (int step = 16; step > 0; step >>= 1)
                            if (k + step < 27 && s_excl[k + step] <= t) k += step;
                        label[cpts[s_cstart[k] + (t - s_excl[k])]] = current;
                    }
                }
            } else if (lane == 0) {
                label[i] = oc;
            }
            __syncthreads();
        }
    }
    if (lane == 0) params[c].n_labels = current;
}

__device__ inline unsigned long long enc_double(double d) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
inline double dec_double(unsigned long long e) {
    const unsigned long long b = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
    double d;
    memcpy(&d, &b, sizeof(d));
    return d;
}

__global__ __launch_bounds__(kBlock) void cluster_stats_kernel(CloudView in, const ClassParams *__restrict__ params,
                                                               const int32_t *__restrict__ counts,
                                                               const int32_t *__restrict__ fg_order,
                                                               const int32_t *__restrict__ label_all,
                                                               const int32_t *__restrict__ parent_all,
                                                               ClusterStat *__restrict__ stat, int32_t *__restrict__ pt_cluster) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= counts[5]) return;
    int c = 0;
#pragma unroll
    for (int k = 1; k < kNumFgClasses; ++k)
        if (g >= params[k].start) c = k;
    const int start = params[c].start;
    const int j = g - start;
    const int v = find_root(parent_all + start + c, label_all[g]);
    const int oi = fg_order[g];
    pt_cluster[oi] = v;
    ClusterStat *s = stat + start + c + v;
    atomicAdd(&s->count, 1);
    atomicMin(&s->first, j);
    const unsigned long long ex = enc_double(in.x[oi]), ey = enc_double(in.y[oi]), ez = enc_double(in.z[oi]);
    atomicMin(&s->lo[0], ex); atomicMax(&s->hi[0], ex);
    atomicMin(&s->lo[1], ey); atomicMax(&s->hi[1], ey);
    atomicMin(&s->lo[2], ez); atomicMax(&s->hi[2], ez);
}

// ---- BuildGraph edges + histogram (F/SemGraph.cpp:24-62): one block per node row ----
__global__ __launch_bounds__(128) void graph_edges_kernel(const double *__restrict__ boxes, int nb, double edge_dis_th,
                                                          int sub_interval_value, int subinterval,
                                                          double *__restrict__ edge, int32_t *__restrict__ hist) {
    extern __shared__ int s_hist[];
    const int i = blockIdx.x;
    const int cols = 3 * subinterval;
    for (int k = threadIdx.x; k < cols; k += blockDim.x) s_hist[k] = 0;
    __syncthreads();
    const D3 ci{boxes[8 * i], boxes[8 * i + 1], boxes[8 * i + 2]};
    for (int j = threadIdx.x; j < nb; j += blockDim.x) {
        const D3 cj{boxes[8 * j], boxes[8 * j + 1], boxes[8 * j + 2]};
        const double e = sqrt(sq3(sub(ci, cj)));
        double out = -1.0;  // -1 marks "no edge" (a stored 0 is the self edge)
        if (e < edge_dis_th) {
            out = e;
            const int lj = (int)boxes[8 * j + 6];
            const int bin = (int)(e / sub_interval_value);
            if (lj >= 1 && lj <= 3 && bin >= 0 && bin < subinterval) atomicAdd(&s_hist[(lj - 1) * subinterval + bin], 1);
        }
        edge[(size_t)i * nb + j] = out;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < cols; k += blockDim.x) hist[(size_t)i * cols + k] = s_hist[k];
}

// Symmetric eigen-decomposition on the host (N is tens of nodes): Householder tridiagonalisation with
// explicit accumulation of Q, then implicit symmetric QR steps with Wilkinson shifts (Golub & Van Loan
// 8.3), rotations applied to the columns of Q.  Eigenvalues ascending like SelfAdjointEigenSolver.
void jacobi_eig(int n, std::vector<double> &a, std::vector<double> &evals, std::vector<double> &evecs) {
    evecs.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) evecs[(size_t)i * n + i] = 1.0;
    auto A = [&](int i, int j) -> double & { return a[(size_t)i * n + j]; };
    std::vector<double> vt((size_t)n * n, 0.0);  // transposed eigenvector matrix: row k = eigenvector k
    for (int i = 0; i < n; ++i) vt[(size_t)i * n + i] = 1.0;
    std::vector<double> v((size_t)n), pvec((size_t)n), w((size_t)n), sacc_r((size_t)n);
    for (int k = 0; k + 2 < n; ++k) {
        double tail2 = 0;
        for (int i = k + 2; i < n; ++i) tail2 += A(i, k) * A(i, k);
        if (tail2 == 0.0) continue;  // column already tridiagonal
        const double x0 = A(k + 1, k);
        const double alpha = -std::copysign(std::sqrt(x0 * x0 + tail2), x0);
        for (int i = k + 1; i < n; ++i) v[(size_t)i] = A(i, k);
        v[(size_t)k + 1] -= alpha;
        double vn2 = 0;
        for (int i = k + 1; i < n; ++i) vn2 += v[(size_t)i] * v[(size_t)i];
        if (vn2 == 0.0) continue;
        const double beta = 2.0 / vn2;
        for (int i = k + 1; i < n; ++i) {
            double sacc = 0;
            for (int j = k + 1; j < n; ++j) sacc += A(i, j) * v[(size_t)j];
            pvec[(size_t)i] = beta * sacc;
        }
        double kk = 0;
        for (int i = k + 1; i < n; ++i) kk += v[(size_t)i] * pvec[(size_t)i];
        kk *= 0.5 * beta;
        for (int i = k + 1; i < n; ++i) w[(size_t)i] = pvec[(size_t)i] - kk * v[(size_t)i];
        for (int i = k + 1; i < n; ++i)
            for (int j = k + 1; j < n; ++j) A(i, j) -= v[(size_t)i] * w[(size_t)j] + w[(size_t)i] * v[(size_t)j];
        A(k + 1, k) = A(k, k + 1) = alpha;
        for (int i = k + 2; i < n; ++i) A(i, k) = A(k, i) = 0.0;
        std::fill(sacc_r.begin(), sacc_r.end(), 0.0);
        for (int j = k + 1; j < n; ++j) {
            const double vj = v[(size_t)j];
            const double *row = &vt[(size_t)j * n];
            for (int r = 0; r < n; ++r) sacc_r[(size_t)r] += row[r] * vj;
        }
        for (int j = k + 1; j < n; ++j) {
            const double vj = beta * v[(size_t)j];
            double *row = &vt[(size_t)j * n];
            for (int r = 0; r < n; ++r) row[r] -= sacc_r[(size_t)r] * vj;
        }
    }
    std::vector<double> d((size_t)n), e((size_t)std::max(n, 1), 0.0);
    for (int i = 0; i < n; ++i) d[(size_t)i] = A(i, i);
    for (int i = 0; i + 1 < n; ++i) e[(size_t)i] = A(i + 1, i);
    const double eps = 2.220446049250313e-16;
    for (int hi = n - 1; hi > 0; --hi) {
        for (int iter = 0; iter < 200; ++iter) {
            if (std::fabs(e[(size_t)hi - 1]) <= eps * (std::fabs(d[(size_t)hi - 1]) + std::fabs(d[(size_t)hi]))) {
                e[(size_t)hi - 1] = 0.0;
                break;
            }
            int lo = hi - 1;
            while (lo > 0 && std::fabs(e[(size_t)lo - 1]) > eps * (std::fabs(d[(size_t)lo - 1]) + std::fabs(d[(size_t)lo]))) --lo;
            const double delta = 0.5 * (d[(size_t)hi - 1] - d[(size_t)hi]);
            const double eh = e[(size_t)hi - 1];
            const double denom = delta + std::copysign(std::hypot(delta, eh), delta == 0.0 ? 1.0 : delta);
            const double mu = d[(size_t)hi] - eh * eh / denom;
            double x = d[(size_t)lo] - mu, z = e[(size_t)lo];
            for (int k = lo; k < hi; ++k) {
                const double r = std::hypot(x, z);
                const double c = r == 0.0 ? 1.0 : x / r, sn = r == 0.0 ? 0.0 : -z / r;
                if (k > lo) e[(size_t)k - 1] = r;
                const double aa = d[(size_t)k], bb = e[(size_t)k], gg = d[(size_t)k + 1];
                d[(size_t)k] = c * c * aa - 2.0 * c * sn * bb + sn * sn * gg;
                d[(size_t)k + 1] = sn * sn * aa + 2.0 * c * sn * bb + c * c * gg;
                e[(size_t)k] = c * sn * (aa - gg) + (c * c - sn * sn) * bb;
                if (k < hi - 1) {
                    z = -sn * e[(size_t)k + 1];
                    e[(size_t)k + 1] = c * e[(size_t)k + 1];
                    x = e[(size_t)k];
                }
                double *rk = &vt[(size_t)k * n], *rk1 = &vt[(size_t)(k + 1) * n];
                for (int rr = 0; rr < n; ++rr) {
                    const double qk = rk[rr], qk1 = rk1[rr];
                    rk[rr] = c * qk - sn * qk1;
                    rk1[rr] = sn * qk + c * qk1;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) A(i, i) = d[(size_t)i];
    for (int r = 0; r < n; ++r)
        for (int k = 0; k < n; ++k) evecs[(size_t)r * n + k] = vt[(size_t)k * n + r];
    evals.resize((size_t)n);
    for (int i = 0; i < n; ++i) evals[(size_t)i] = A(i, i);
    // ascending order, columns permuted alongside (SelfAdjointEigenSolver convention)
    std::vector<int> idx((size_t)n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return evals[(size_t)x] < evals[(size_t)y]; });
    std::vector<double> ev2((size_t)n), vec2((size_t)n * n);
    for (int c = 0; c < n; ++c) {
        ev2[(size_t)c] = evals[(size_t)idx[(size_t)c]];
        for (int r = 0; r < n; ++r) vec2[(size_t)r * n + c] = evecs[(size_t)r * n + idx[(size_t)c]];
    }
    evals.swap(ev2);
    evecs.swap(vec2);
}

template <class T>
bool galloc(T **p, size_t count) {
    void *q = nullptr;
    if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) return false;
    *p = (T *)q;
    return true;
}

int gfail(std::string *err, int code, const char *msg) {
    if (err) *err = msg;
    return code;
}

}  // namespace

int graph_scratch_alloc(GraphScratch &g, int N) {
    g.N = N;
    const size_t n = (size_t)N;
    bool ok = galloc(&g.cat_blockcnt, (n / kBlock + 2) * 8) && galloc(&g.fg_order, n) && galloc(&g.fg_all, n) &&
              galloc(&g.bg_all, n) && galloc(&g.counts, 8) && galloc(&g.params, kNumFgClasses) && galloc(&g.cell3, n * 4) &&
              galloc(&g.cellslot, n) && galloc(&g.cell_pts, n) && galloc(&g.table, n * 4 + 16) && galloc(&g.label, n) &&
              galloc(&g.parent, n + 16) && galloc(&g.stat, n + 16) && galloc(&g.pt_cluster, n) &&
              galloc(&g.boxes_dev, (size_t)kMaxNodes * 8) && galloc(&g.edge_dev, (size_t)kMaxNodes * kMaxNodes) &&
              galloc(&g.hist_dev, (size_t)kMaxNodes * kMaxHistCols);
    if (ok) ok = hipHostMalloc((void **)&g.pinned, 1024, hipHostMallocDefault) == hipSuccess;
    return ok ? SGS_OK : SGS_ENOMEM;
}

void graph_scratch_free(GraphScratch &g) {
    void *ps[] = {g.cat_blockcnt, g.fg_order, g.fg_all, g.bg_all, g.counts, g.params, g.cell3, g.cellslot, g.cell_pts,
                  g.table, g.label, g.parent, g.stat, g.pt_cluster, g.boxes_dev, g.edge_dev, g.hist_dev};
    for (void *p : ps)
        if (p) hipFree(p);
    if (g.pinned) hipHostFree(g.pinned);
    g = GraphScratch();
}

// All cluster kernels + the small async read-back of counts / class parameters; no host sync.
int graph_cluster_enqueue(hipStream_t s, GraphScratch &g, CloudView in, const int32_t *n_ptr, int n_upper,
                          const sgs_config &cfg, std::string *err) {
    g.enqueued_upper = n_upper;
    if (n_upper <= 0) return SGS_OK;
    if (!(cfg.deltaA > 0) || !(cfg.deltaR > 0) || !(cfg.deltaP > 0)) return gfail(err, SGS_EINVAL, "cluster deltas must be positive");
    const int nb = (n_upper + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(route_count_kernel, dim3(nb), dim3(kBlock), 0, s, in, n_ptr, g.cat_blockcnt, g.pt_cluster);
    hipLaunchKernelGGL(route_totals_kernel, dim3(1), dim3(kBlock), 0, s, g.cat_blockcnt, nb, g.counts, g.params);
    hipLaunchKernelGGL(route_scatter_kernel, dim3(nb), dim3(kBlock), 0, s, in, n_ptr, g.cat_blockcnt, g.params, g.fg_order,
                       g.fg_all, g.bg_all);
    hipLaunchKernelGGL(stat_init_kernel, dim3((n_upper + 16 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, g.stat, n_upper + 16);
    ClusterConsts K;
    K.dA = cfg.deltaA; K.dR = cfg.deltaR; K.dP = cfg.deltaP;
    // F/SemanticCluster.hpp:46-47 and .cpp:38-39, evaluated on the host in the reference's expression order
    const double min_az = -24.8 * M_PI / 180, max_az = 2 * M_PI / 180;
    K.min_az = min_az;
    K.width = int(round(360 / cfg.deltaP));
    K.height = int(((max_az - min_az) * 180 / M_PI) / cfg.deltaA) + 1;
    hipLaunchKernelGGL(cluster_class_kernel, dim3(kNumFgClasses), dim3(kBlock), 0, s, in, g.params, g.fg_order, K, g.cell3,
                       g.cellslot, g.cell_pts, g.table, g.label, g.parent);
    hipLaunchKernelGGL(cluster_stats_kernel, dim3(nb), dim3(kBlock), 0, s, in, g.params, g.counts, g.fg_order, g.label,
                       g.parent, g.stat, g.pt_cluster);
    // counts (8 ints) and params (5 structs) land in one pinned block
    if (hipMemcpyAsync(g.pinned, g.counts, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(g.pinned + 64, g.params, sizeof(ClassParams) * kNumFgClasses, hipMemcpyDeviceToHost, s) != hipSuccess)
        return gfail(err, SGS_EHIP, "cluster: device error");
    return SGS_OK;
}

// Waits for the cluster kernels, reads the per-cluster statistics back in one copy and emits the boxes
// in the reference's order.
int graph_cluster_finish(hipStream_t s, GraphScratch &g, const int32_t *n_ptr, std::vector<double> *boxes, int *n_boxes,
                         int32_t *point_cluster_host, int *n_fg, int *n_bg, std::string *err) {
    if (boxes) boxes->clear();
    if (n_boxes) *n_boxes = 0;
    if (n_fg) *n_fg = 0;
    if (n_bg) *n_bg = 0;
    if (g.enqueued_upper <= 0) return SGS_OK;
    if (hipStreamSynchronize(s) != hipSuccess) return gfail(err, SGS_EHIP, "cluster: device error");
    int32_t counts[8];
    memcpy(counts, g.pinned, sizeof(counts));
    memcpy(g.params_host, g.pinned + 64, sizeof(ClassParams) * kNumFgClasses);
    if (n_fg) *n_fg = counts[5];
    if (n_bg) *n_bg = counts[6];
    int total = 0;
    static const int kClassLabel[kNumFgClasses] = {1, 1, 1, 2, 3};  // :311-322
    const size_t nstat = (size_t)counts[5] + kNumFgClasses + 1;
    std::vector<ClusterStat> &all = g.stat_host;
    all.resize(nstat);
    if (counts[5] > 0 && (hipMemcpyAsync(all.data(), g.stat, sizeof(ClusterStat) * nstat, hipMemcpyDeviceToHost, s) != hipSuccess ||
                          hipStreamSynchronize(s) != hipSuccess))
        return gfail(err, SGS_EHIP, "cluster: device error");
    for (int c = 0; c < kNumFgClasses; ++c) {
        const ClassParams &p = g.params_host[c];
        if (p.n <= 0 || p.n_labels <= 0) continue;
        const ClusterStat *st = all.data() + p.start + c + 1;
        // replay histcounts[...] insertion (:278-304): keys enter in order of first appearance
        std::vector<int> ids;
        for (int v = 1; v <= p.n_labels; ++v)
            if (st[(size_t)v - 1].count > 0) ids.push_back(v);
        std::sort(ids.begin(), ids.end(), [&](int a, int b) { return st[(size_t)a - 1].first < st[(size_t)b - 1].first; });
        std::unordered_map<int, int> hist;
        for (int v : ids) hist[v] = v;
        for (const auto &kv : hist) {
            const ClusterStat &q = st[(size_t)kv.first - 1];
            if (q.count < 10) continue;
            double lo[3], hi[3];
            for (int a = 0; a < 3; ++a) { lo[a] = dec_double(q.lo[a]); hi[a] = dec_double(q.hi[a]); }
            const double dim[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
            const int lab = kClassLabel[c];
            if (lab == 1) {
                if (dim[0] < 0.5 || dim[1] < 0.5 || dim[2] < 0.3) continue;
            } else if (dim[2] < 0.3) {
                continue;
            }
            if (boxes) {
                const double b[8] = {(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2, dim[0], dim[1], dim[2], (double)lab, (double)q.count};
                boxes->insert(boxes->end(), b, b + 8);
            }
            ++total;
        }
    }
    if (n_boxes) *n_boxes = total;
    if (point_cluster_host) {
        int n = 0;
        if (hipMemcpyAsync(&n, n_ptr, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return gfail(err, SGS_EHIP, "cluster: device error");
        if (n > 0 && (hipMemcpyAsync(point_cluster_host, g.pt_cluster, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess ||
                      hipStreamSynchronize(s) != hipSuccess))
            return gfail(err, SGS_EHIP, "cluster: device error");
    }
    return SGS_OK;
}

int graph_cluster(hipStream_t s, GraphScratch &g, CloudView in, const int32_t *n_ptr, int n_upper, const sgs_config &cfg,
                  std::vector<double> *boxes, int *n_boxes, int32_t *point_cluster_host, int *n_fg, int *n_bg,
                  std::string *err) {
    int r = graph_cluster_enqueue(s, g, in, n_ptr, n_upper, cfg, err);
    if (r) return r;
    return graph_cluster_finish(s, g, n_ptr, boxes, n_boxes, point_cluster_host, n_fg, n_bg, err);
}

int graph_build(hipStream_t s, GraphScratch &g, const double *boxes, int nb, double edge_dis_th, int subinterval, int dim,
                double subgraph_edge_th, float *desc, double *edge_matrix, double *eigenvalues, int32_t *edges,
                double *edge_value, double *edge_weights, int edge_cap, int *n_edges, int32_t *tri_counts, double *triangles,
                int tri_cap, int *n_triangles, int32_t *class_counts3, std::string *err) {
    if (n_edges) *n_edges = 0;
    if (n_triangles) *n_triangles = 0;
    if (class_counts3) class_counts3[0] = class_counts3[1] = class_counts3[2] = 0;
    if (nb == 0) return SGS_OK;
    if (nb > kMaxNodes) return gfail(err, SGS_ECAPACITY, "too many graph nodes");
    if (subinterval < 1 || 3 * subinterval > kMaxHistCols || dim < 0) return gfail(err, SGS_EINVAL, "bad graph parameters");
    const int sub_interval_value = int(edge_dis_th / subinterval);  // F/SemGraph.cpp:16 (integer!)
    if (sub_interval_value < 1) return gfail(err, SGS_EINVAL, "edge_dis_th / subinterval truncates to 0");
    const int cols = 3 * subinterval;
    if (hipMemcpyAsync(g.boxes_dev, boxes, sizeof(double) * 8 * (size_t)nb, hipMemcpyHostToDevice, s) != hipSuccess)
        return gfail(err, SGS_EHIP, "graph: upload failed");
    hipLaunchKernelGGL(graph_edges_kernel, dim3(nb), dim3(128), sizeof(int) * (size_t)cols, s, g.boxes_dev, nb, edge_dis_th,
                       sub_interval_value, subinterval, g.edge_dev, g.hist_dev);
    std::vector<double> E((size_t)nb * nb);
    std::vector<int32_t> H((size_t)nb * cols);
    if (hipMemcpyAsync(E.data(), g.edge_dev, sizeof(double) * E.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(H.data(), g.hist_dev, sizeof(int32_t) * H.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return gfail(err, SGS_EHIP, "graph: device error");
    // adjacency / edge matrix / edge lists (F/SemGraph.cpp:36-47)
    std::vector<double> Adj((size_t)nb * nb, 0.0);
    int ne = 0;
    for (int i = 0; i < nb; ++i) {
        const int li = (int)boxes[8 * (size_t)i + 6];
        if (class_counts3 && li >= 1 && li <= 3) class_counts3[li - 1]++;
        for (int j = 0; j < nb; ++j) {
            double &e = E[(size_t)i * nb + j];
            if (e >= 0) {
                Adj[(size_t)i * nb + j] = 1;
                if (j >= i + 1) {
                    if (ne < edge_cap) {
                        if (edges) { edges[2 * ne] = i; edges[2 * ne + 1] = j; }
                        if (edge_value) edge_value[ne] = e;
                        if (edge_weights) edge_weights[ne] = (edge_dis_th - e) / edge_dis_th;
                    }
                    ++ne;
                }
            } else {
                e = 0.0;
            }
        }
    }
    if (n_edges) *n_edges = ne;
    if (edge_matrix) memcpy(edge_matrix, E.data(), sizeof(double) * E.size());
    // spectral half (MatrixDecomposing :215-240): |eigvec| columns of the largest |eigenvalue|s
    std::vector<double> evals, evecs;
    jacobi_eig(nb, Adj, evals, evecs);
    if (eigenvalues) memcpy(eigenvalues, evals.data(), sizeof(double) * (size_t)nb);
    if (desc) {
        std::vector<float> absev((size_t)nb);
        for (int i = 0; i < nb; ++i) absev[(size_t)i] = (float)std::fabs(evals[(size_t)i]);
        std::vector<int> idx((size_t)nb);
        std::iota(idx.begin(), idx.end(), 0);
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return absev[(size_t)a] > absev[(size_t)b]; });
        const int D = cols + dim;
        const int iter = std::min(dim, nb);
        std::vector<double> G((size_t)dim);
        for (int i = 0; i < nb; ++i) {
            double s2 = 0;
            for (int k = 0; k < cols; ++k) s2 += (double)H[(size_t)i * cols + k] * (double)H[(size_t)i * cols + k];
            const double nrm = std::sqrt(s2);
            for (int k = 0; k < cols; ++k) desc[(size_t)i * D + k] = (float)((double)H[(size_t)i * cols + k] / nrm);
            double sum = 0;
            for (int k = 0; k < dim; ++k) {
                G[(size_t)k] = k < iter ? std::fabs(evecs[(size_t)i * nb + idx[(size_t)k]]) : 0.0;
                sum += G[(size_t)k];
            }
            for (int k = 0; k < dim; ++k) desc[(size_t)i * D + cols + k] = (float)(G[(size_t)k] / sum);
        }
    }
    // sub-triangles (:80-103)
    int nt = 0;
    for (int i = 0; i < nb; ++i) {
        std::vector<int> ind;
        for (int m = 0; m < nb; ++m) {
            const double e = E[(size_t)i * nb + m];
            if (e < subgraph_edge_th && e != 0) ind.push_back(m);
        }
        int cnt = 0;
        if (ind.size() > 2) {
            auto ctr = [&](int k) { return D3{boxes[8 * (size_t)k], boxes[8 * (size_t)k + 1], boxes[8 * (size_t)k + 2]}; };
            for (size_t m = 0; m + 1 < ind.size(); ++m)
                for (size_t q = m + 1; q < ind.size(); ++q) {
                    float t[3];
                    t[0] = (float)std::sqrt(sq3(sub(ctr(i), ctr(ind[m]))));
                    t[1] = (float)std::sqrt(sq3(sub(ctr(i), ctr(ind[q]))));
                    t[2] = (float)std::sqrt(sq3(sub(ctr(ind[m]), ctr(ind[q]))));
                    std::sort(t, t + 3);
                    if (triangles && nt < tri_cap) { triangles[3 * (size_t)nt] = t[0]; triangles[3 * (size_t)nt + 1] = t[1]; triangles[3 * (size_t)nt + 2] = t[2]; }
                    ++nt;
                    ++cnt;
                }
        }
        if (tri_counts) tri_counts[i] = cnt;
    }
    if (n_triangles) *n_triangles = nt;
    if (ne > edge_cap && (edges || edge_value || edge_weights)) return gfail(err, SGS_ECAPACITY, "edge capacity too small");
    if (nt > tri_cap && triangles) return gfail(err, SGS_ECAPACITY, "triangle capacity too small");
    return SGS_OK;
}

int graph_finish_scan(hipStream_t s, GraphScratch &g, CloudView in, const int32_t *n_ptr, const sgs_config &cfg,
                      std::vector<double> *boxes, int *n_boxes, std::string *err) {
    (void)in;
    int r = graph_cluster_finish(s, g, n_ptr, boxes, n_boxes, nullptr, nullptr, nullptr, err);
    if (r) return r;
    const int nb = *n_boxes;
    if (nb == 0) return SGS_OK;
    g.last_desc.assign((size_t)nb * (3 * cfg.subinterval + cfg.graph_node_dimension), 0.f);
    int ne = 0, nt = 0;
    return graph_build(s, g, boxes->data(), nb, cfg.edge_dis_th, cfg.subinterval, cfg.graph_node_dimension,
                       cfg.subgraph_edge_th, g.last_desc.data(), nullptr, nullptr, nullptr, nullptr, nullptr, 0, &ne, nullptr,
                       nullptr, 0, &nt, nullptr, err);
}

}  // namespace sgs
