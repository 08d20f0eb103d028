// topk.hpp -- wavefront-wide (64 lanes) exact top-k kept in registers.
//
// The list of the k best (key, id) pairs seen so far is striped over the wave: slot s = e*64 + lane
// (e < KPL) holds the s-th smallest pair; k <= 64*KPL.  A new pair is offered by every lane; lanes
// whose pair beats the current k-th entry are drained one at a time (ballot + readlane) and inserted
// with a one-slot shift done by `__shfl_up` -- no LDS, no divergence inside the insertion.
// Order: (key ascending, id ascending), the canonical tie rule of the library.
#pragma once
#include "common.hpp"

namespace vdb {

__device__ __forceinline__ bool pair_less(uint64_t ka, int64_t ia, uint64_t kb, int64_t ib) {
    return (ka < kb) || (ka == kb && ia < ib);
}

__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src_lane) {
    unsigned lo = __builtin_amdgcn_readlane((unsigned)v, src_lane);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

template <int KPL>
struct WaveTopK {
    static constexpr int kBatchMin = 12;   // accepted pairs per offer() from which the batch merge beats serial insertion
    uint64_t key[KPL];
    int64_t id[KPL];
    int n;             // entries held (wave-uniform)
    int k;             // capacity requested (wave-uniform, <= 64*KPL)
    uint64_t thr_key;  // k-th entry when full
    int64_t thr_id;

    __device__ __forceinline__ void init(int k_) {
        k = k_;
        n = 0;
        thr_key = ~0ull;
        thr_id = INT64_MAX;
#pragma unroll
        for (int e = 0; e < KPL; ++e) {
            key[e] = ~0ull;
            id[e] = -1;
        }
    }

    __device__ __forceinline__ bool beats_threshold(uint64_t nk, int64_t nid) const {
        return n < k || pair_less(nk, nid, thr_key, thr_id);
    }

    // wave-uniform arguments
    __device__ __forceinline__ void insert_uniform(uint64_t nk, int64_t nid) {
        const int lane = threadIdx.x & 63;
        int pos = 0;
#pragma unroll
        for (int e = 0; e < KPL; ++e) {
            const int s = e * 64 + lane;
            const bool lt = (s < n) && pair_less(key[e], id[e], nk, nid);
            pos += __popcll(__ballot(lt));
        }
#pragma unroll
        for (int e = KPL - 1; e >= 0; --e) {
            uint64_t upk = __shfl_up(key[e], 1);
            int64_t upi = __shfl_up(id[e], 1);
            if (e > 0) {
                const uint64_t pk = bcast_u64(key[e - 1], 63);
                const int64_t pi = (int64_t)bcast_u64((uint64_t)id[e - 1], 63);
                if (lane == 0) {
                    upk = pk;
                    upi = pi;
                }
            }
            const int s = e * 64 + lane;
            if (s > pos) {
                key[e] = upk;
                id[e] = upi;
            } else if (s == pos) {
                key[e] = nk;
                id[e] = nid;
            }
        }
        if (n < k) ++n;
        if (n == k) {
            const int ts = k - 1;
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                if (e == (ts >> 6)) {
                    thr_key = bcast_u64(key[e], ts & 63);
                    thr_id = (int64_t)bcast_u64((uint64_t)id[e], ts & 63);
                }
            }
        }
    }

    // compare-exchange of this lane's pair with the pair of lane ^ j: keep the smaller one if keep_min
    __device__ __forceinline__ static void cex_lane(uint64_t &k_, int64_t &i_, int j, bool keep_min) {
        const uint64_t ok = __shfl_xor(k_, j);
        const int64_t oi = __shfl_xor(i_, j);
        const bool other_less = pair_less(ok, oi, k_, i_);
        if (other_less == keep_min) {
            k_ = ok;
            i_ = oi;
        }
    }

    // Batch form of offer() for MANY accepted pairs (large k: the one-at-a-time insertion costs ~40-70 instructions
    // per pair and made the top-k, not the distances, the cost of k >= 100 searches).  The <= 64 offered pairs are
    // sorted across the lanes (bitonic network, 21 compare-exchange steps), reversed, and merged into the sorted list
    // with one half-cleaner step against the list's last register followed by a bitonic merge of the list itself.
    // Unused slots carry the sentinel (~0, INT64_MAX) while this runs; `n` keeps track of the real entries.
    __device__ __forceinline__ void merge_batch(uint64_t bk, int64_t bi, int cnt) {
        const int lane = threadIdx.x & 63;
        // 1. sort the batch ascending over the lanes
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1) {
            const bool asc = (lane & k2) == 0;          // (k2 == 64: every lane ascending)
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) cex_lane(bk, bi, j, ((lane & j) == 0) == asc);
        }
        // 2. reverse it (descending): list ascending ++ batch descending is a bitonic sequence
        bk = __shfl(bk, 63 - lane);
        bi = __shfl(bi, 63 - lane);
        // sentinels in the unused slots of the list
#pragma unroll
        for (int e = 0; e < KPL; ++e)
            if (e * 64 + lane >= n) {
                key[e] = ~0ull;
                id[e] = INT64_MAX;
            }
        // 3. half-cleaner: the virtual second half is (sentinel registers ..., batch); only the last register meets real data
        if (pair_less(bk, bi, key[KPL - 1], id[KPL - 1])) {
            key[KPL - 1] = bk;
            id[KPL - 1] = bi;
        }
        // 4. the list is now bitonic and holds the KPL*64 smallest pairs: bitonic merge, register strides then lane strides
#pragma unroll
        for (int re = KPL >> 1; re > 0; re >>= 1) {
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                if ((e & re) == 0) {
                    const int o = e + re;
                    if (pair_less(key[o], id[o], key[e], id[e])) {
                        const uint64_t tk_ = key[e];
                        const int64_t ti_ = id[e];
                        key[e] = key[o];
                        id[e] = id[o];
                        key[o] = tk_;
                        id[o] = ti_;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 32; j > 0; j >>= 1) {
#pragma unroll
            for (int e = 0; e < KPL; ++e) cex_lane(key[e], id[e], j, (lane & j) == 0);
        }
        // 5. bookkeeping: the list holds min(n + cnt, k) entries; slots past k are dropped
        n = (n + cnt < k) ? n + cnt : k;
#pragma unroll
        for (int e = 0; e < KPL; ++e)
            if (e * 64 + lane >= n) {
                key[e] = ~0ull;
                id[e] = -1;
            }
        if (n == k) {
            const int ts = k - 1;
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                if (e == (ts >> 6)) {
                    thr_key = bcast_u64(key[e], ts & 63);
                    thr_id = (int64_t)bcast_u64((uint64_t)id[e], ts & 63);
                }
            }
        }
    }

    // every lane offers one pair (valid = false -> nothing offered)
    __device__ __forceinline__ void offer(uint64_t nk, int64_t nid, bool valid) {
        const bool want = valid && beats_threshold(nk, nid);
        uint64_t mask = __ballot(want);
        const int cnt = __popcll(mask);
        if (cnt >= kBatchMin) {
            merge_batch(want ? nk : ~0ull, want ? nid : INT64_MAX, cnt);
            return;
        }
        while (mask) {
            const int src = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            const uint64_t bk = bcast_u64(nk, src);
            const int64_t bi = (int64_t)bcast_u64((uint64_t)nid, src);
            if (beats_threshold(bk, bi)) insert_uniform(bk, bi);
        }
    }
};

}  // namespace vdb
