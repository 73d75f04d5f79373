// ORACLE (test infrastructure): behavioural restatement of tsl::robin_map v1.0.1 as used by the
// reference (pinned at 3rdparty/tsl_robin/tsl_robin.cmake:24; its source is NOT in /root/reference,
// so this is the published robin-hood algorithm restated: power-of-two growth, max load 0.5,
// linear probing with robin-hood displacement, backward-shift deletion).  What leaks into SG-SLAM's
// results is (1) iteration order = ascending bucket index (F/Preprocessing.cpp:87-91,
// F/VoxelHashMap.cpp:136) and (2) erase-while-iterating skipping the element shifted into the
// current bucket (F/VoxelHashMap.cpp:179-185).  PARITY UNPINNED (no reference tests exist).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace sgo {

struct Key3 {
    int32_t x, y, z;
    bool operator==(const Key3 &o) const { return x == o.x && y == o.y && z == o.z; }
};

// C/Coreutils.h:40-45 and F/Preprocessing.cpp:41-46 (identical): 20-bit masked XOR hash on uint32.
inline uint32_t voxel_hash(const Key3 &k) {
    const uint32_t a = (uint32_t)k.x, b = (uint32_t)k.y, c = (uint32_t)k.z;
    return ((1u << 20) - 1u) & (a * 73856093u ^ b * 19349663u ^ c * 83492791u);
}

struct RobinStats {
    int64_t max_dist = 0;        // largest distance-from-ideal-bucket ever stored
    int64_t grow_by_probe = 0;   // times the probe-length heuristic requested a growth
    int64_t rehashes = 0;
    int64_t wrapped = 0;         // elements sitting before their ideal bucket (wrapped past the end)
};

template <class V>
class RobinMap {
public:
    struct Bucket {
        int32_t dist = -1;  // -1 == empty
        Key3 key{0, 0, 0};
        V value{};
    };
    // probe_limit: v1.0.x sets the grow flag when a displaced element's distance exceeds
    // DIST_FROM_IDEAL_BUCKET_LIMIT (4096 in 1.0.x as recalled; 0.6.x used 128 + load>=0.15).
    explicit RobinMap(int64_t probe_limit = 4096) : probe_limit_(probe_limit) {}

    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    size_t bucket_count() const { return buckets_.size(); }
    const std::vector<Bucket> &buckets() const { return buckets_; }
    std::vector<Bucket> &buckets_mut() { return buckets_; }
    const RobinStats &stats() const { return stats_; }

    void clear() {
        for (auto &b : buckets_) b.dist = -1;
        size_ = 0;
        grow_next_ = false;
    }
    // reserve(n): rehash(ceil(n / max_load_factor))
    void reserve(size_t n) {
        size_t want = (size_t)std::ceil((float)n / 0.5f);
        size_t need = (size_t)std::ceil((float)size_ / 0.5f);
        if (need > want) want = need;
        rehash_to(round_pow2(want));
    }
    // returns bucket index or -1
    int64_t find(const Key3 &k) const {
        if (buckets_.empty()) return -1;
        size_t ib = voxel_hash(k) & mask_;
        int32_t d = 0;
        while (d <= buckets_[ib].dist) {
            if (buckets_[ib].key == k) return (int64_t)ib;
            ib = (ib + 1) & mask_;
            ++d;
        }
        return -1;
    }
    bool contains(const Key3 &k) const { return find(k) >= 0; }
    V &value_at(int64_t ib) { return buckets_[(size_t)ib].value; }
    const V &value_at(int64_t ib) const { return buckets_[(size_t)ib].value; }

    // insert if absent; returns (bucket index, inserted)
    std::pair<int64_t, bool> insert(const Key3 &k, V v) {
        size_t ib = buckets_.empty() ? 0 : (voxel_hash(k) & mask_);
        int32_t d = 0;
        if (!buckets_.empty()) {
            while (d <= buckets_[ib].dist) {
                if (buckets_[ib].key == k) return {(int64_t)ib, false};
                ib = (ib + 1) & mask_;
                ++d;
            }
        }
        if (grow_next_ || size_ >= load_threshold_) {
            rehash_to(buckets_.empty() ? 2 : buckets_.size() * 2);
            grow_next_ = false;
            ib = voxel_hash(k) & mask_;
            d = 0;
            while (d <= buckets_[ib].dist) {
                ib = (ib + 1) & mask_;
                ++d;
            }
        }
        place(ib, d, k, std::move(v), true);  // the newcomer always lands in bucket ib
        ++size_;
        return {(int64_t)ib, true};
    }
    // backward-shift deletion; returns true if erased
    bool erase(const Key3 &k) {
        int64_t f = find(k);
        if (f < 0) return false;
        size_t prev = (size_t)f;
        buckets_[prev].dist = -1;
        --size_;
        size_t ib = (prev + 1) & mask_;
        while (buckets_[ib].dist > 0) {
            buckets_[prev] = std::move(buckets_[ib]);
            buckets_[prev].dist -= 1;
            buckets_[ib].dist = -1;
            prev = ib;
            ib = (ib + 1) & mask_;
        }
        return true;
    }
    void compute_wrapped() {
        stats_.wrapped = 0;
        for (size_t i = 0; i < buckets_.size(); ++i)
            if (buckets_[i].dist >= 0 && (size_t)buckets_[i].dist > i) ++stats_.wrapped;
    }

private:
    static size_t round_pow2(size_t v) {
        if (v == 0) return 0;
        size_t p = 1;
        while (p < v) p <<= 1;
        return p;
    }
    void rehash_to(size_t nb) {
        std::vector<Bucket> old;
        old.swap(buckets_);
        buckets_.assign(nb, Bucket{});
        mask_ = nb ? nb - 1 : 0;
        load_threshold_ = (size_t)((float)nb * 0.5f);
        ++stats_.rehashes;
        for (auto &b : old) {
            if (b.dist < 0) continue;
            size_t ib = voxel_hash(b.key) & mask_;
            int32_t d = 0;
            // insert_value_on_rehash: robin-hood without key comparison
            while (d <= buckets_[ib].dist) {
                ib = (ib + 1) & mask_;
                ++d;
            }
            place(ib, d, b.key, std::move(b.value), false);
        }
    }
    void place(size_t ib, int32_t d, Key3 k, V v, bool may_flag) {
        while (true) {
            Bucket &b = buckets_[ib];
            if (b.dist < 0) {
                b.dist = d; b.key = k; b.value = std::move(v);
                if (d > stats_.max_dist) stats_.max_dist = d;
                return;
            }
            if (d > b.dist) {
                if (may_flag && d > probe_limit_) {
                    grow_next_ = true;
                    ++stats_.grow_by_probe;
                }
                std::swap(d, b.dist);
                std::swap(k, b.key);
                std::swap(v, b.value);
                if (b.dist > stats_.max_dist) stats_.max_dist = b.dist;
            }
            ib = (ib + 1) & mask_;
            ++d;
        }
    }
    std::vector<Bucket> buckets_;
    size_t mask_ = 0, size_ = 0, load_threshold_ = 0;
    bool grow_next_ = false;
    int64_t probe_limit_;
    RobinStats stats_;
};

}  // namespace sgo
