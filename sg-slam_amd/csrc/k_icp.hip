// Registration hot loop: VoxelHashMap::GetCorrespondences (F/VoxelHashMap.cpp:51-131) fused with the
// AlignClouds reduction (F/Registration.cpp:82-109), then the 6x6 solve / SE3 update / convergence
// test of RegisterFrameSemantic (F/Registration.cpp:111-112,147-177) in a one-block kernel, so a whole
// Gauss-Newton loop runs without a host round trip.
//
// Mapping: ONE WAVEFRONT PER QUERY (grid-stride over queries).  Lanes 0..26 probe the 27 neighbour
// voxels in the reference's x-outer / z-inner order (one 16-B slot-header load per probe gives key +
// count).  The occupied voxels are compacted across the wave with ds_permute (cross-lane through the
// LDS crossbar, no LDS allocation), and their points - voxel-local fp32 (x,y,z,label) records stored
// contiguously per voxel - are read 64 candidates at a time as coalesced 16-B loads straight into
// registers (each candidate is used once; staging it in LDS would only add a round trip).
// Nearest neighbour = min over (fp32 squared distance, visiting ordinal) packed in one u64, reduced by
// wave shuffles; first-wins tie-break like the reference's strict '<'.
// The robust weighted normal equations need only 16 moment sums (J = [I | -hat(s)] is fixed by s):
//   W = sum w, S = sum w s, P = sum w s s^T (6), bt = sum w ws r, br = sum w ws (s x r)
// accumulated in fp64 per wave, combined per block through LDS in fixed order, written as per-block
// partials and folded by the solve kernel in fixed order: bit-reproducible run to run, no atomics.
// This is gather/latency bound integer + fp work; there is no dense contraction, hence no MFMA.
#include <climits>

#include "sgs_device.h"

namespace sgs {

namespace {

struct QueryResult {
    bool found;      // a nearest neighbour exists among the 27 voxels
    D3 target;       // its world coordinates (voxel origin + fp32 offset)
    int label;       // its label
    int candidates;  // resident points examined
    int probes;      // slot headers inspected
};

// whole wave cooperates on one query point w (world frame); every lane returns the same result
__device__ inline QueryResult wave_nearest(const MapView &m, D3 w, int lane) {
    const K3 kc = voxel_of(w.x, w.y, w.z, m.voxel);
    int slot = -1;
    uint32_t cnt = 0;
    int probes = 0;
    if (lane < 27) {
        const K3 kn{kc.x + lane / 9 - 1, kc.y + (lane / 3) % 3 - 1, kc.z + lane % 3 - 1};
        if (key_in_range(kn)) {
            const unsigned long long pk = pack_key(kn);
            uint32_t s = map_hash(pk) & m.mask;
            for (uint32_t step = 0; step <= m.mask; ++step) {
                const ulonglong2 raw = reinterpret_cast<const ulonglong2 *>(m.slots)[s];
                ++probes;
                if (raw.x == pk) {
                    slot = (int)s;
                    cnt = (uint32_t)(raw.y & 0xFFFFFFFFull);
                    break;
                }
                if (raw.x == kEmptyKey) break;
                s = (s + 1u) & m.mask;
            }
        }
    }
    const bool occ = slot >= 0 && cnt > 0;
    const unsigned long long occmask = __ballot(occ);
    const int nf = __popcll(occmask);
    // stable partition of the lanes: occupied voxels first, in visiting order
    const int before = __popcll(occmask & ((1ull << lane) - 1ull));
    const int dst = occ ? before : nf + (lane - before);
    const int c_slot = __builtin_amdgcn_ds_permute(dst << 2, slot);
    const int c_meta = __builtin_amdgcn_ds_permute(dst << 2, (int)(cnt | ((uint32_t)lane << 8)));
    // candidates: compacted voxel k, point p  <->  flat index c = k*cap + p
    unsigned long long best = ~0ull;
    float bx = 0.f, by = 0.f, bz = 0.f;
    int blabel = 0;
    int ncand = 0;
    const int total = nf * m.cap;
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + lane;
        const int k = c / m.cap;
        const int p = c - k * m.cap;
        const int kk = k < nf ? k : 0;
        const int es = __shfl(c_slot, kk);
        const int em = __shfl(c_meta, kk);
        const int ecnt = em & 0xFF, v = em >> 8;
        if (k < nf && p < ecnt) {
            const float4 q = m.pts[(size_t)es * m.cap + p];
            const double ox = (double)(kc.x + v / 9 - 1) * m.voxel;
            const double oy = (double)(kc.y + (v / 3) % 3 - 1) * m.voxel;
            const double oz = (double)(kc.z + v % 3 - 1) * m.voxel;
            const float dx = q.x - (float)(w.x - ox), dy = q.y - (float)(w.y - oy), dz = q.z - (float)(w.z - oz);
            const float d2 = dx * dx + (dy * dy + dz * dz);
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(uint32_t)(v * m.cap + p);
            if (key < best) {
                best = key;
                bx = q.x; by = q.y; bz = q.z;
                blabel = __float_as_int(q.w);
            }
            ++ncand;
        }
    }
    unsigned long long wbest = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(wbest, o);
        wbest = other < wbest ? other : wbest;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ncand += __shfl_xor(ncand, o);
        probes += __shfl_xor(probes, o);
    }
    QueryResult r;
    r.candidates = ncand;
    r.probes = probes;
    r.found = wbest != ~0ull;
    r.label = 0;
    r.target = D3{0, 0, 0};
    if (r.found) {
        const int winner = __ffsll((long long)__ballot(best == wbest)) - 1;
        const float fx = __shfl(bx, winner), fy = __shfl(by, winner), fz = __shfl(bz, winner);
        r.label = __shfl(blabel, winner);
        const int v = (int)(wbest & 0xFFFFFFFFull) / m.cap;
        r.target = D3{(double)(kc.x + v / 9 - 1) * m.voxel + (double)fx, (double)(kc.y + (v / 3) % 3 - 1) * m.voxel + (double)fy,
                      (double)(kc.z + v % 3 - 1) * m.voxel + (double)fz};
    }
    return r;
}

__global__ __launch_bounds__(kIcpThreads) void icp_accumulate_kernel(MapView m, CloudView src,
                                                                     const int32_t *__restrict__ n_ptr, DevState *st,
                                                                     double *__restrict__ partials) {
    if (st->icp_done) return;
    const int n = *n_ptr;
    const Rigid T = st->T_total;
    const double max_corr = st->max_corr, th = st->kernel_th;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = kIcpThreads / 64;
    double acc[kIcpSums];
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) acc[k] = 0.0;
    for (int q = blockIdx.x * waves_per_block + wave; q < n; q += gridDim.x * waves_per_block) {
        const D3 s = apply(T, D3{src.x[q], src.y[q], src.z[q]});
        const int label = src.l[q];
        const QueryResult r = wave_nearest(m, s, lane);
        acc[17] += (double)r.candidates;
        acc[18] += (double)r.probes;
        if (!r.found) continue;
        const D3 res = sub(s, r.target);
        const double r2 = sq3(res);
        if (!(sqrt(r2) < max_corr)) continue;
        const double wgt = (th * th) / ((th + r2) * (th + r2));
        const double ws = wgt * semantic_weight(label);
        acc[0] += wgt;
        acc[1] += wgt * s.x; acc[2] += wgt * s.y; acc[3] += wgt * s.z;
        acc[4] += wgt * s.x * s.x; acc[5] += wgt * s.y * s.y; acc[6] += wgt * s.z * s.z;
        acc[7] += wgt * s.x * s.y; acc[8] += wgt * s.x * s.z; acc[9] += wgt * s.y * s.z;
        acc[10] += ws * res.x; acc[11] += ws * res.y; acc[12] += ws * res.z;
        const D3 sxr = crs(s, res);
        acc[13] += ws * sxr.x; acc[14] += ws * sxr.y; acc[15] += ws * sxr.z;
        acc[16] += 1.0;
    }
    __shared__ double red[kIcpThreads / 64][kIcpSums];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kIcpSums; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < kIcpSums) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kIcpThreads / 64; ++w) sum += red[w][threadIdx.x];
        partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sum;  // [scalar][block]: the fold reads rows
    }
}

// fixed-order fold of the per-block partials ([scalar][block] layout): each wave folds whole rows
// with independent coalesced loads + a shuffle tree, so the result is bit-reproducible
__device__ inline void fold_partials(const double *__restrict__ partials, int nblocks, double *out /*shared [kIcpSums]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int sc = wave; sc < kIcpSums; sc += nwaves) {
        double sum = 0.0;
        for (int b = lane; b < nblocks; b += 64) sum += partials[(size_t)sc * nblocks + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if (lane == 0) out[sc] = sum;
    }
    __syncthreads();
}

__device__ inline void normal_equations(const double *S, double A[6][6], double b[6]) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) A[i][j] = 0.0;
    const double W = S[0], sx = S[1], sy = S[2], sz = S[3];
    const double xx = S[4], yy = S[5], zz = S[6], xy = S[7], xz = S[8], yz = S[9];
    A[0][0] = A[1][1] = A[2][2] = W;
    // sum w * (-hat(s)) and its transpose
    A[0][4] = sz;  A[0][5] = -sy;
    A[1][3] = -sz; A[1][5] = sx;
    A[2][3] = sy;  A[2][4] = -sx;
    A[4][0] = sz;  A[5][0] = -sy;
    A[3][1] = -sz; A[5][1] = sx;
    A[3][2] = sy;  A[4][2] = -sx;
    // sum w * (|s|^2 I - s s^T)
    A[3][3] = yy + zz; A[4][4] = xx + zz; A[5][5] = xx + yy;
    A[3][4] = A[4][3] = -xy;
    A[3][5] = A[5][3] = -xz;
    A[4][5] = A[5][4] = -yz;
    for (int i = 0; i < 6; ++i) b[i] = S[10 + i];
}

// mailbox: 4 ints in host-pinned, device-mapped memory {iterations finished, done, -, -}; the host
// polls it instead of issuing a copy + stream sync per chunk of iterations
__global__ __launch_bounds__(256) void icp_solve_kernel(DevState *st, const double *__restrict__ partials, int nblocks,
                                                        int32_t *mailbox) {
    if (st->icp_done) return;
    __shared__ double S[kIcpSums];
    fold_partials(partials, nblocks, S);
    if (threadIdx.x != 0) return;
    double A[6][6], b[6], rhs[6], x[6];
    normal_equations(S, A, b);
#pragma unroll
    for (int i = 0; i < 6; ++i) rhs[i] = -b[i];
    solve6_static(A, rhs, x);
    const Rigid est = rigid_exp_small(x);
    const Rigid T_icp = compose(est, st->T_icp);
    st->T_icp = T_icp;
    st->T_total = compose(T_icp, st->guess);
    st->icp_corr += (long long)S[16];
    st->icp_cand += (long long)S[17];
    st->icp_probe += (long long)S[18];
    const int it = st->icp_iter + 1;
    st->icp_iter = it;
    // estimation.log().norm() (F/Registration.cpp:169): log(exp(x)) == x for |omega| < pi, so the
    // 6-vector norm of the update itself is the termination quantity
    double nn = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) nn += x[i] * x[i];
    const int done = (sqrt(nn) < 0.0001 || it >= st->icp_max_iter) ? 1 : 0;
    if (done) st->icp_done = 1;
    if (mailbox) {
        __hip_atomic_store(&mailbox[0], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox[1], done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void icp_finish_kernel(DevState *st, const double *__restrict__ partials, int nblocks,
                                                         double *__restrict__ out28) {
    __shared__ double S[kIcpSums];
    fold_partials(partials, nblocks, S);
    if (threadIdx.x != 0) return;
    double A[6][6], b[6];
    normal_equations(S, A, b);
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) out28[k++] = A[i][j];
    for (int i = 0; i < 6; ++i) out28[k++] = b[i];
    out28[27] = S[16];
    st->icp_cand += (long long)S[17];
    st->icp_probe += (long long)S[18];
}

__global__ void icp_init_kernel(DevState *st, Rigid guess, double max_corr, double kernel_th, int max_iter) {
    st->guess = guess;
    st->T_icp = rigid_identity();
    st->T_total = guess;
    st->max_corr = max_corr;
    st->kernel_th = kernel_th;
    st->icp_iter = 0;
    st->icp_done = 0;
    st->icp_max_iter = max_iter;
    st->icp_corr = 0;
    st->icp_cand = 0;
    st->icp_probe = 0;
}

// per-query correspondences for sgs_map_get_correspondences (points given in world frame)
__global__ __launch_bounds__(256) void correspondences_kernel(MapView m, CloudView src, int n, double max_dist,
                                                              double *__restrict__ tgt_xyzl, int32_t *__restrict__ found) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int q = blockIdx.x * 4 + wave; q < n; q += gridDim.x * 4) {
        const D3 s{src.x[q], src.y[q], src.z[q]};
        const QueryResult r = wave_nearest(m, s, lane);
        bool ok = false;
        if (r.found) ok = sqrt(sq3(sub(r.target, s))) < max_dist;
        if (lane == 0) {
            found[q] = ok ? 1 : 0;
            tgt_xyzl[4 * (size_t)q] = r.target.x; tgt_xyzl[4 * (size_t)q + 1] = r.target.y;
            tgt_xyzl[4 * (size_t)q + 2] = r.target.z; tgt_xyzl[4 * (size_t)q + 3] = (double)r.label;
        }
    }
}

}  // namespace

void launch_icp_init(hipStream_t s, DevState *st, const Rigid &guess, double max_corr, double kernel_th, int max_iter,
                     int) {
    hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, s, st, guess, max_corr, kernel_th, max_iter);
}

void launch_icp_iteration(hipStream_t s, MapView m, CloudView src, const int32_t *n_ptr, DevState *st, double *partials,
                          bool solve, int32_t *mailbox) {
    hipLaunchKernelGGL(icp_accumulate_kernel, dim3(kIcpBlocks), dim3(kIcpThreads), 0, s, m, src, n_ptr, st, partials);
    if (solve) hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(256), 0, s, st, partials, kIcpBlocks, mailbox);
}

void launch_icp_iteration_solve_only(hipStream_t s, DevState *st, double *partials, int32_t *mailbox) {
    hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(256), 0, s, st, partials, kIcpBlocks, mailbox);
}

void launch_linear_system_finish(hipStream_t s, DevState *st, double *partials, double *out28_dev) {
    hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(256), 0, s, st, partials, kIcpBlocks, out28_dev);
}

void launch_correspondences(hipStream_t s, MapView m, CloudView src, int n, double max_dist, double *tgt_xyzl,
                            int32_t *found) {
    if (n <= 0) return;
    int nb = (n + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(correspondences_kernel, dim3(nb), dim3(256), 0, s, m, src, n, max_dist, tgt_xyzl, found);
}

}  // namespace sgs
