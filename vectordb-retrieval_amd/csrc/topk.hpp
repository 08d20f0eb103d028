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

    // every lane offers one pair (valid = false -> nothing offered)
    __device__ __forceinline__ void offer(uint64_t nk, int64_t nid, bool valid) {
        uint64_t mask = __ballot(valid && beats_threshold(nk, nid));
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
