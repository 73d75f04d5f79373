// GPU-resident voxel hash map (F/VoxelHashMap.{hpp,cpp}): open addressing with linear probing over
// 16-byte slot headers {packed key, count, pending-list head}; slot s owns the storage block
// pts[s*cap .. s*cap+cap) of 32-byte point records (fp64 world coordinates + label), so a find is one 16-B load
// and a voxel's points are one contiguous coalesced read.
//
// AddPoints semantics to preserve (F/VoxelHashMap.cpp:164-176): points are inserted sequentially in
// input order and a voxel keeps at most `cap` points first-come.  Parallel version:
//   add kernel     every point finds-or-claims its slot (atomicCAS on the key), pushes itself on the
//                  slot's pending list (atomicExch on head) and counts itself in the upper 24 bits
//                  of the slot's count word
//   append kernel  every point walks its voxel's pending list ONCE to find its rank among the
//                  pending points by input index; rank r goes to position cnt + r if that is below
//                  cap - exactly the points the sequential loop would have stored, in its order
//   sweep kernel   (always follows) folds the pending count into the count and clears the list head.
// Prune (F/VoxelHashMap.cpp:178-186): a sweep over the slot headers that tombstones voxels whose
// oldest point is farther than max_distance from the origin (pure predicate; the reference's
// erase-while-iterating skip quirk is layout dependent and not reproduced).
#include <climits>

#include "sgs_device.h"

namespace sgs {

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void map_init_kernel(MapView m) {
    const uint32_t cap = m.mask + 1u;
    ulonglong2 *t = reinterpret_cast<ulonglong2 *>(m.slots);
    // key = EMPTY, cnt = 0, head = -1
    const ulonglong2 empty = make_ulonglong2(kEmptyKey, 0xFFFFFFFF00000000ull);
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < cap; s += gridDim.x * kBlock) t[s] = empty;
}

__global__ void map_reset_counters_kernel(DevState *st) {
    st->map_live = 0; st->map_tomb = 0; st->map_points = 0;
}

__global__ __launch_bounds__(kBlock) void map_add_kernel(MapView m, CloudView in, const int32_t *__restrict__ n_ptr,
                                                         const Rigid *__restrict__ T_dev, int32_t *__restrict__ slot_of,
                                                         int32_t *__restrict__ next, CloudView world, DevState *st) {
    if (m.skip && *m.skip) return;
    const int n = *n_ptr;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    D3 p{in.x[i], in.y[i], in.z[i]};
    if (T_dev) p = apply(*T_dev, p);
    world.x[i] = p.x; world.y[i] = p.y; world.z[i] = p.z;
    const K3 k = voxel_of(p.x, p.y, p.z, m.voxel);
    if (!key_in_range(k)) {
        atomicOr(&st->err_flags, ERR_KEY_RANGE);
        slot_of[i] = -1;
        return;
    }
    if (m.shard_world > 1 && !shard_stores(k, m.shard_log2, m.shard_world, m.shard_rank)) {
        slot_of[i] = -1;  // another rank's voxel, outside this rank's halo
        return;
    }
    const unsigned long long pk = pack_key(k);
    uint32_t s = map_hash(pk) & m.mask;
    int found = -1;
    for (uint32_t step = 0; step <= m.mask; ++step) {
        unsigned long long cur = m.slots[s].key;
        if (cur == kEmptyKey) {
            cur = atomicCAS(&m.slots[s].key, kEmptyKey, pk);
            if (cur == kEmptyKey) {
                found = (int)s;
                break;
            }
        }
        if (cur == pk) { found = (int)s; break; }
        s = (s + 1u) & m.mask;
    }
    if (found < 0) {
        atomicOr(&st->err_flags, ERR_MAP_FULL);
        slot_of[i] = -1;
        return;
    }
    slot_of[i] = found;
    next[i] = atomicExch(&m.slots[found].head, i);
    atomicAdd(&m.slots[found].cnt, 1u << 8);  // pending points of this batch (the stored count is <= 255)
}

__global__ __launch_bounds__(kBlock) void map_append_kernel(MapView m, CloudView in, const int32_t *__restrict__ n_ptr,
                                                            const int32_t *__restrict__ slot_of,
                                                            const int32_t *__restrict__ next, CloudView world,
                                                            DevState *st) {
    if (m.skip && *m.skip) return;
    const int n = *n_ptr;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    if (s < 0) return;
    // header: stable during this kernel (the sweep that follows finalises count and head)
    const ulonglong2 raw = reinterpret_cast<const ulonglong2 *>(m.slots)[s];
    const uint32_t cntword = (uint32_t)(raw.y & 0xFFFFFFFFull);
    const int cnt0 = (int)(cntword & 0xFFu);
    if (cnt0 >= m.cap) return;
    const int head = (int)(uint32_t)(raw.y >> 32);
    int rank = 0;
    for (int j = head; j >= 0; j = next[j]) rank += (j < i) ? 1 : 0;
    const int pos = cnt0 + rank;
    if (pos >= m.cap) return;
    m.pts[(size_t)s * m.cap + pos] = MapPoint{world.x[i], world.y[i], world.z[i], in.l[i], 0};
    (void)st;
}

// Sweep over all slot headers: optional prune + per-block census (live voxels, tombstones, points);
// map_census_kernel folds the per-block numbers into DevState so no contended atomics are needed.
__global__ __launch_bounds__(kBlock) void map_prune_kernel(MapView m, const Rigid *__restrict__ T_dev, D3 origin_host,
                                                           double max_d2, int do_prune, int4 *__restrict__ blockstat) {
    if (m.skip && *m.skip) return;
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    int live = 0, tomb = 0, pts = 0;
    if (s <= m.mask) {
        const ulonglong2 raw = reinterpret_cast<const ulonglong2 *>(m.slots)[s];
        const unsigned long long key = raw.x;
        const uint32_t cntword = (uint32_t)(raw.y & 0xFFFFFFFFull);
        int cnt = (int)(cntword & 0xFFu);
        const bool valid = key != kTombKey && key != kEmptyKey;
        const int pending = (int)(cntword >> 8);
        if (pending) {  // a batch of map_add: stored = min(cap, stored + pending), list consumed
            cnt = min(m.cap, cnt + pending);
            if (valid) reinterpret_cast<ulonglong2 *>(m.slots)[s] = make_ulonglong2(key, 0xFFFFFFFF00000000ull | (unsigned long long)(uint32_t)cnt);
        }
        bool kill = false;
        if (valid && do_prune) {
            const D3 o = T_dev ? T_dev->t : origin_host;
            const MapPoint f = m.pts[(size_t)s * m.cap];  // points.front() (F/VoxelHashMap.cpp:181)
            kill = sq3(sub(D3{f.x, f.y, f.z}, o)) > max_d2;
        }
        live = (valid && !kill) ? 1 : 0;
        tomb = (key == kTombKey || kill) ? 1 : 0;
        pts = live ? cnt : 0;
        if (kill) {
            // key = TOMB, cnt = 0, head = -1 in one 16-byte store
            reinterpret_cast<ulonglong2 *>(m.slots)[s] = make_ulonglong2(kTombKey, 0xFFFFFFFF00000000ull);
        }
    }
    __shared__ int red[3][kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_down(live, o);
        tomb += __shfl_down(tomb, o);
        pts += __shfl_down(pts, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = live; red[1][threadIdx.x >> 6] = tomb; red[2][threadIdx.x >> 6] = pts;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        blockstat[blockIdx.x] = make_int4(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3],
                                          red[2][0] + red[2][1] + red[2][2] + red[2][3], 0);
}

constexpr int kCensusBlock = 1024;  // one workgroup folds all per-block numbers: wide, so that the loads of a large map's sweep are few rounds
__global__ __launch_bounds__(kCensusBlock) void map_census_kernel(const int4 *__restrict__ blockstat, int nblocks, DevState *st,
                                                                  HostMirror *mirror, const int32_t *__restrict__ skip) {
    if (skip && *skip) return;
    int live = 0, tomb = 0, pts = 0;
#pragma unroll 4
    for (int b = threadIdx.x; b < nblocks; b += kCensusBlock) {
        const int4 v = blockstat[b];
        live += v.x; tomb += v.y; pts += v.z;
    }
    __shared__ int red[3][kCensusBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_down(live, o);
        tomb += __shfl_down(tomb, o);
        pts += __shfl_down(pts, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = live; red[1][threadIdx.x >> 6] = tomb; red[2][threadIdx.x >> 6] = pts;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int l = 0, t = 0, p = 0;
#pragma unroll
        for (int w = 0; w < kCensusBlock / 64; ++w) {
            l += red[0][w]; t += red[1][w]; p += red[2][w];
        }
        st->map_live = l;
        st->map_tomb = t;
        st->map_points = p;
        if (mirror) {
            mirror->map_live = l;
            mirror->map_tomb = t;
            mirror->map_points = p;
            mirror->map_err_flags = st->err_flags;
            mirror->t_census_end = wall_clock64();
            const int32_t seq = st->census_seq + 1;  // counted in device memory: the host-mapped mirror is write-only for kernels
            st->census_seq = seq;
            __hip_atomic_store(&mirror->census_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(kBlock) void map_rehash_kernel(MapView from, MapView to, DevState *st) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s > from.mask) return;
    const MapSlot e = from.slots[s];
    if (e.key == kEmptyKey || e.key == kTombKey) return;
    uint32_t d = map_hash(e.key) & to.mask;
    for (uint32_t step = 0; step <= to.mask; ++step) {
        if (atomicCAS(&to.slots[d].key, kEmptyKey, e.key) == kEmptyKey) break;
        d = (d + 1u) & to.mask;
    }
    to.slots[d].cnt = e.cnt;
    for (uint32_t j = 0; j < e.cnt; ++j) to.pts[(size_t)d * to.cap + j] = from.pts[(size_t)s * from.cap + j];
    (void)st;
}

__global__ void map_after_rehash_kernel(DevState *st) { st->map_tomb = 0; }

}  // namespace

void launch_map_init(hipStream_t s, MapView m) {
    const uint32_t cap = m.mask + 1u;
    int nb = (int)((cap + kBlock - 1) / kBlock);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(map_init_kernel, dim3(nb), dim3(kBlock), 0, s, m);
}

void launch_map_clear(hipStream_t s, MapView m, DevState *st) {
    launch_map_init(s, m);
    hipLaunchKernelGGL(map_reset_counters_kernel, dim3(1), dim3(1), 0, s, st);
}

void launch_map_add(hipStream_t s, MapView m, CloudView in, const int32_t *n_ptr, int n_upper, const Rigid *T_dev,
                    int32_t *slot_of, int32_t *next, CloudView world, DevState *st) {
    if (n_upper <= 0) return;
    const int nb = (n_upper + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(map_add_kernel, dim3(nb), dim3(kBlock), 0, s, m, in, n_ptr, T_dev, slot_of, next, world, st);
    hipLaunchKernelGGL(map_append_kernel, dim3(nb), dim3(kBlock), 0, s, m, in, n_ptr, slot_of, next, world, st);
}

void launch_map_sweep(hipStream_t s, MapView m, const Rigid *T_dev, const double *origin3_host, double max_dist,
                      int do_prune, int4 *blockstat, DevState *st, HostMirror *mirror) {
    const uint32_t cap = m.mask + 1u;
    D3 o{0, 0, 0};
    if (origin3_host) o = D3{origin3_host[0], origin3_host[1], origin3_host[2]};
    const int nb = (int)((cap + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(map_prune_kernel, dim3(nb), dim3(kBlock), 0, s, m, T_dev, o, max_dist * max_dist, do_prune,
                       blockstat);
    hipLaunchKernelGGL(map_census_kernel, dim3(1), dim3(kCensusBlock), 0, s, blockstat, nb, st, mirror, m.skip);
}

void launch_map_rehash(hipStream_t s, MapView from, MapView to, DevState *st) {
    launch_map_init(s, to);
    const uint32_t cap = from.mask + 1u;
    hipLaunchKernelGGL(map_rehash_kernel, dim3((cap + kBlock - 1) / kBlock), dim3(kBlock), 0, s, from, to, st);
    hipLaunchKernelGGL(map_after_rehash_kernel, dim3(1), dim3(1), 0, s, st);
}

}  // namespace sgs
