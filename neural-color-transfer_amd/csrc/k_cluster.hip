// k_cluster.hip — C1 (k-means labels of the coarsest S features) and K1 (per-cluster kNN graph in Lab).
// Reference: ColorTransfer::clusterFeastures ColorTransfer.cpp:355-395 -> cvflann KMeansIndex::computeClustering
// (Flann/kmeans_index.h:700-880, chooseCentersRandom :108-135, L2 distance Flann/dist.h:153-181);
// ColorTransfer::findKnns/getClusters/findSubKNNs/sortMergeComputeWeight ColorTransfer.cpp:397-423,273-353,136-195,60-110
// (nanoflann KD-trees, one per cluster, built and queried on the host under OpenMP in the reference).
//
// MI355X design:
//  * k-means: tiny (<= 63x63 points x 512-d, k=10, <= 11 Lloyd steps) and sequential between steps: four small kernels per
//    step (centres: one thread per (cluster, dim) summing members in point order in fp64 exactly like the reference loop;
//    distances: one thread per (point, centre) in the reference's float accumulation order; relabel; step end), all
//    gated by a device-side `done` flag so the host never synchronises (bit-identical labels to the oracle).
//  * kNN: exact search on a colour grid instead of KD-trees. Lab comes from 8-bit images, so the squared distance is an
//    exact small integer: all (cluster, pixel) memberships are radix-sorted by (cluster, 8-unit colour cell), a query visits
//    cells in Chebyshev rings around its own cell with an integer pre-filter against the current 9th-best and stops when no
//    unvisited cell can hold a closer or tying point; only survivors pay for the fp64 sqrt that defines the final
//    (dist, id) order (= the reference's cmpDist order; ties are first-come in the reference's KD-tree).
#include "nct_internal.h"
#include "nct_device.h"
#include "nct_detmath.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>   // rocPRIM directly (no CUB-compatibility layer)
#include <algorithm>

// ================================================================= C1: k-means
__device__ float l2_fd(const float* __restrict__ a, const double* __restrict__ b, int n) {
    float result = 0.f;
    int i = 0;
    // sixteen elements per trip, their loads issued together (the loop below, one group of four at a time, ran one L1 round trip per group: 37 us for the 19 360 (pixel,
    // centre) pairs of a 700 x 700 pair's conv5_1 map). Same operations in the same order: the groups of four are summed and added exactly as below.
    if ((((size_t)a | (size_t)b) & 15) == 0) {
        for (; i + 15 < n; i += 16) {
            float4 av[4]; double2 bv[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) av[u] = reinterpret_cast<const float4*>(a + i)[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = reinterpret_cast<const double2*>(b + i)[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d0 = (float)((double)av[u].x - bv[2 * u].x), d1 = (float)((double)av[u].y - bv[2 * u].y);
                const float d2 = (float)((double)av[u].z - bv[2 * u + 1].x), d3 = (float)((double)av[u].w - bv[2 * u + 1].y);
                result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
    }
    for (; i + 3 < n; i += 4) {
        const float d0 = (float)((double)a[i] - b[i]), d1 = (float)((double)a[i + 1] - b[i + 1]);
        const float d2 = (float)((double)a[i + 2] - b[i + 2]), d3 = (float)((double)a[i + 3] - b[i + 3]);
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; i < n; ++i) { const float d0 = (float)((double)a[i] - b[i]); result += d0 * d0; }
    return result;
}
__device__ float l2_ff(const float* __restrict__ a, const float* __restrict__ b, int n) {
    float result = 0.f;
    int i = 0;
    if ((((size_t)a | (size_t)b) & 15) == 0) {                   // sixteen elements per trip as in l2_fd: the same groups of four in the same order
        for (; i + 15 < n; i += 16) {
            float4 av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = reinterpret_cast<const float4*>(a + i)[u]; bv[u] = reinterpret_cast<const float4*>(b + i)[u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d0 = av[u].x - bv[u].x, d1 = av[u].y - bv[u].y, d2 = av[u].z - bv[u].z, d3 = av[u].w - bv[u].w;
                result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
    }
    for (; i + 3 < n; i += 4) {
        const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; i < n; ++i) { const float d0 = a[i] - b[i]; result += d0 * d0; }
    return result;
}

constexpr int KM_MAXK = 16;
constexpr int KM_LDS_PERM = 4096;
struct KMState { int cidx[KM_MAXK]; int count[KM_MAXK]; unsigned radius[KM_MAXK]; int nc, changed, done; };

// centre selection (chooseCentersRandom with a SplitMix64 Fisher-Yates permutation). One 256-thread workgroup: SplitMix64 is counter
// based, so all swap partners are computed in parallel; only the n dependent swaps run in one thread (LDS, ~50 ns each); the
// duplicate test of a candidate against the centres chosen so far runs one centre per thread (each distance in the reference's
// sequential float order).
__device__ __forceinline__ uint64_t sm64_at(uint64_t seed, uint64_t draw /*0-based*/) {
    uint64_t z = seed + (draw + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_km_init(const float* __restrict__ feat, int n, int C, int K, uint64_t seed, int* __restrict__ perm, KMState* __restrict__ st,
                                                 int* __restrict__ labels) {
    __shared__ int s_perm[KM_LDS_PERM]; __shared__ __attribute__((aligned(16))) int s_j[KM_LDS_PERM];
    __shared__ int s_cand, s_dup, s_stop;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) labels[i] = 0;
    if (tid == 0) {
        st->nc = 0; st->changed = 0; st->done = n < K ? 1 : 0;
        for (int i = 0; i < KM_MAXK; ++i) { st->count[i] = 0; st->radius[i] = 0u; }
    }
    if (n < K) return;
    const bool lds = n <= KM_LDS_PERM;
    int* pm = lds ? s_perm : perm;
    for (int i = tid; i < n; i += 256) pm[i] = i;
    if (lds) {
        // step i (n-1 down to 1) uses draw n-1-i: j = draw mod (i+1)
        for (int i = tid + 1; i < n; i += 256) {
            const uint64_t z = sm64_at(seed, (uint64_t)(n - 1 - i));
            const uint32_t d = (uint32_t)(i + 1);
            // z mod d from 32-bit remainders (d < 2^16): z = hi*2^32 + lo  =>  ((hi mod d) * (2^32 mod d) + lo mod d) mod d
            const uint32_t hi = (uint32_t)(z >> 32), lo = (uint32_t)z;
            const uint32_t c = ((0xFFFFFFFFu % d) + 1u) % d;
            s_j[i] = (int)(((hi % d) * c + lo % d) % d);
        }
    }
    __syncthreads();
    // Only the HEAD of the permutation is ever read (K candidates + one per duplicate). The n dependent swaps in one thread cost 210 us at n = 1 936; the value that ends
    // up at position l is found without performing them: walk the swaps backwards in time (i = 1 ... n - 1: the last one executed first) and follow where the element at l
    // came from — lane l of wave 0 for positions 0 ... 63, every step one uniform LDS read and two compares. Same permutation, entry for entry. The full shuffle runs
    // only if more than 64 positions are consumed (more than 64 - K duplicates: a flat image), or for maps beyond the LDS list (n > 4096).
    __shared__ int s_head[64];
    constexpr int HEAD = 64;
    if (lds) {
        if (tid < HEAD) {
            int pos = tid;
            auto back = [&](int i, int j) { pos = pos == i ? j : (pos == j ? i : pos); };
            int i = 1;
            for (; i < n && (i & 15); ++i) back(i, s_j[i]);
            for (; i + 16 <= n; i += 16) {                       // sixteen swap partners per LDS trip (one read per step left the loop at LDS latency: 120 us)
                int4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = reinterpret_cast<const int4*>(s_j + i)[u];
#pragma unroll
                for (int u = 0; u < 4; ++u) { back(i + 4 * u, q[u].x); back(i + 4 * u + 1, q[u].y); back(i + 4 * u + 2, q[u].z); back(i + 4 * u + 3, q[u].w); }
            }
            for (; i < n; ++i) back(i, s_j[i]);
            s_head[tid] = pos;                                   // pm[x] = x before the first swap
        }
    } else if (tid == 0) {
        for (int i = n - 1; i > 0; --i) { const int j = (int)(sm64_at(seed, (uint64_t)(n - 1 - i)) % (uint64_t)(i + 1)); const int t = pm[i]; pm[i] = pm[j]; pm[j] = t; }
    }
    __syncthreads();
    bool shuffled = !lds;                                        // thread 0 only: pm holds the complete permutation
    // The duplicate test compares the candidate's vector with every centre chosen so far (l2_ff: a serial chain per pair). From global memory that chain ran one round trip
    // per group of four (~20 us per candidate, 190 of the kernel's 195 us at C = 512): the candidate's row and the chosen centres' rows are staged in LDS instead (C <= 512,
    // K <= 10; rows padded by four floats), the chains read LDS. Same operations in the same order.
    constexpr int CMAX = 512, KLDS = 10, CST = CMAX + 4;
    __shared__ __attribute__((aligned(16))) float s_cent[KLDS * CST]; __shared__ __attribute__((aligned(16))) float s_row[CST];
    const bool rows_lds = C <= CMAX && K <= KLDS && (C & 3) == 0;
    // first K candidates of the permutation that are not duplicates (squared distance < 1e-16) of an earlier centre
    int pos = 0, nc = 0;
    for (int index = 0; index < K; ++index) {
        bool out = false;
        while (true) {
            if (tid == 0) {
                s_stop = pos >= n ? 1 : 0; s_dup = 0;
                if (pos < n) {
                    if (!shuffled && pos >= HEAD) { for (int i = n - 1; i > 0; --i) { const int j = s_j[i]; const int t = pm[i]; pm[i] = pm[j]; pm[j] = t; } shuffled = true; }
                    s_cand = shuffled ? pm[pos] : s_head[pos]; st->cidx[index] = s_cand;
                }
            }
            __syncthreads();
            if (s_stop) { out = true; break; }
            ++pos;
            if (rows_lds) {
                for (int e = tid; e < (C >> 2); e += 256) reinterpret_cast<float4*>(s_row)[e] = reinterpret_cast<const float4*>(feat + (size_t)s_cand * C)[e];
                __syncthreads();
                if (tid < index && l2_ff(s_row, s_cent + tid * CST, C) < 1e-16) s_dup = 1;
            } else if (tid < index && l2_ff(feat + (size_t)s_cand * C, feat + (size_t)st->cidx[tid] * C, C) < 1e-16) s_dup = 1;
            __syncthreads();
            const bool dup = s_dup != 0;
            __syncthreads();
            if (!dup) break;
        }
        if (out) break;
        if (rows_lds) { for (int e = tid; e < (C >> 2); e += 256) reinterpret_cast<float4*>(s_cent + index * CST)[e] = reinterpret_cast<const float4*>(s_row)[e]; __syncthreads(); }
        nc = index + 1;
    }
    if (tid == 0) { st->nc = nc; if (nc < K) st->done = 1; }           // root cannot be split: one label
}
// first = 1: centres = the chosen points; else: fp64 mean of the members, accumulated in point order (kmeans_index.h:771-785)
__global__ void k_km_centres(const float* __restrict__ feat, int n, int C, int K, const int* __restrict__ labels, KMState* __restrict__ st, double* __restrict__ dc, int first) {
    if (st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * C) return;
    const int c = i / C, k = i - c * C;
    if (first) { dc[i] = (double)feat[(size_t)st->cidx[c] * C + k]; return; }
    // members are added in point order (that order defines the fp64 sum); the loads of eight points are issued together
    double s = 0.0;
    int p = 0;
    for (; p + 8 <= n; p += 8) {
        int lb[8]; float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { lb[u] = labels[p + u]; f[u] = feat[(size_t)(p + u) * C + k]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (lb[u] == c) s += (double)f[u];
    }
    for (; p < n; ++p) if (labels[p] == c) s += (double)feat[(size_t)p * C + k];
    dc[i] = s / (double)st->count[c];
}
// same sums for C a multiple of 256 (every workgroup then belongs to ONE cluster): wave 0 first compacts the cluster's members into an
// ascending list in LDS, then every thread adds only the members — 10x fewer loop trips than scanning all points per (cluster, dim)
__global__ __launch_bounds__(256) void k_km_centres_list(const float* __restrict__ feat, int n, int C, int K, const int* __restrict__ labels, KMState* __restrict__ st,
                                                         double* __restrict__ dc) {
    if (st->done) return;
    __shared__ int s_list[KM_LDS_PERM];
    __shared__ int s_cnt;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = (blockIdx.x * 256) / C, k = i - c * C;
    if (threadIdx.x < 64) {
        int base = 0;
        for (int p0 = 0; p0 < n; p0 += 512) {                 // eight chunks of 64 labels per trip (the loads together), compacted chunk by chunk: the list stays ascending
            int lb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int p = p0 + 64 * u + threadIdx.x; lb[u] = p < n ? labels[p] : -1; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = lb[u] == c;
                const unsigned long long bal = __ballot(in);
                if (in) s_list[base + __popcll(bal & ((1ull << threadIdx.x) - 1ull))] = p0 + 64 * u + threadIdx.x;
                base += __popcll(bal);
            }
        }
        if (threadIdx.x == 0) s_cnt = base;
    }
    __syncthreads();
    const int m = s_cnt;
    double s = 0.0;
    int t = 0;
    for (; t + 32 <= m; t += 32) {                     // the loads of 32 members in flight together; added in list order
        float f[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) f[u] = feat[(size_t)s_list[t + u] * C + k];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += (double)f[u];
    }
    for (; t + 8 <= m; t += 8) {
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = feat[(size_t)s_list[t + u] * C + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)f[u];
    }
    for (; t < m; ++t) s += (double)feat[(size_t)s_list[t] * C + k];
    dc[i] = s / (double)st->count[c];
}
__global__ void k_km_dist(const float* __restrict__ feat, int n, int C, int K, const double* __restrict__ dc, const KMState* __restrict__ st, float* __restrict__ dist) {
    if (st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int p = i / K, c = i - p * K;
    dist[i] = l2_fd(feat + (size_t)p * C, dc + (size_t)c * C, C);
}
// The same distances for C = 512 (the pipeline's conv5_1 map) from LDS: a workgroup stages the K centres (fp64) and the rows of KM_DP pixels once — the general kernel
// re-reads a 2 KB row per thread through an L1 that 26 rows + 40 KB of centres do not fit (34 us for 1 936 pixels) — then thread (pixel, centre) runs the same chain
// (l2_fd: groups of four, in order). Rows are padded against bank conflicts (threads of one pixel read the same feature address: a broadcast; their centres differ).
constexpr int KM_DP = 24, KM_DC = 512, KM_FPAD = 4, KM_CPAD = 2;
__global__ __launch_bounds__(256) void k_km_dist512(const float* __restrict__ feat, int n, int K, const double* __restrict__ dc, const KMState* __restrict__ st, float* __restrict__ dist) {
    if (st->done) return;
    extern __shared__ double s_km[];
    double* s_c = s_km;                                                   // [K][KM_DC + KM_CPAD]
    float* s_f = reinterpret_cast<float*>(s_c + (size_t)K * (KM_DC + KM_CPAD));      // [KM_DP][KM_DC + KM_FPAD]
    const int p0 = blockIdx.x * KM_DP, np = min(KM_DP, n - p0);
    for (int e = threadIdx.x; e < K * (KM_DC / 2); e += 256) {
        const int c = e / (KM_DC / 2), j = e - c * (KM_DC / 2);
        reinterpret_cast<double2*>(s_c + (size_t)c * (KM_DC + KM_CPAD))[j] = reinterpret_cast<const double2*>(dc + (size_t)c * KM_DC)[j];
    }
    for (int e = threadIdx.x; e < np * (KM_DC / 4); e += 256) {
        const int r = e / (KM_DC / 4), j = e - r * (KM_DC / 4);
        reinterpret_cast<float4*>(s_f + (size_t)r * (KM_DC + KM_FPAD))[j] = reinterpret_cast<const float4*>(feat + (size_t)(p0 + r) * KM_DC)[j];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= np * K) return;
    const int r = t / K, c = t - r * K;
    const float* a = s_f + (size_t)r * (KM_DC + KM_FPAD);
    const double* b = s_c + (size_t)c * (KM_DC + KM_CPAD);
    float result = 0.f;
#pragma unroll 4
    for (int i = 0; i < KM_DC; i += 4) {
        const float4 av = *reinterpret_cast<const float4*>(a + i);
        const double2 b0 = *reinterpret_cast<const double2*>(b + i), b1 = *reinterpret_cast<const double2*>(b + i + 2);
        const float d0 = (float)((double)av.x - b0.x), d1 = (float)((double)av.y - b0.y), d2 = (float)((double)av.z - b1.x), d3 = (float)((double)av.w - b1.y);
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    dist[(size_t)(p0 + r) * K + c] = result;
}
// first-minimum assignment (`if (sq_dist > new_sq_dist)`), radius = max, counts, change flag. The cluster statistics are order-free (integer sums, a max of non-negative
// floats as unsigned): a workgroup folds them in LDS and issues one global atomic per touched cluster instead of three per pixel (1 936 pixels hammered ten addresses: 16 us)
__global__ __launch_bounds__(256) void k_km_relabel(int n, int K, const float* __restrict__ dist, int* __restrict__ labels, KMState* __restrict__ st, int first) {
    if (st->done) return;
    __shared__ int s_dcount[KM_MAXK]; __shared__ unsigned s_rad[KM_MAXK]; __shared__ int s_changed;
    if (threadIdx.x < KM_MAXK) { s_dcount[threadIdx.x] = 0; s_rad[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) s_changed = 0;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        float sq = dist[(size_t)p * K]; int b = 0;
        for (int j = 1; j < K; ++j) { const float nsq = dist[(size_t)p * K + j]; if (sq > nsq) { b = j; sq = nsq; } }
        atomicMax(&s_rad[b], __float_as_uint(sq));
        if (first) { labels[p] = b; atomicAdd(&s_dcount[b], 1); }
        else {
            const int old = labels[p];
            if (b != old) { atomicSub(&s_dcount[old], 1); atomicAdd(&s_dcount[b], 1); labels[p] = b; s_changed = 1; }
        }
    }
    __syncthreads();
    if (threadIdx.x < K) {
        if (s_rad[threadIdx.x]) atomicMax(&st->radius[threadIdx.x], s_rad[threadIdx.x]);
        if (s_dcount[threadIdx.x]) atomicAdd(&st->count[threadIdx.x], s_dcount[threadIdx.x]);
    }
    if (threadIdx.x == 0 && s_changed) st->changed = 1;
}
// end of a Lloyd step: empty-cluster repair (kmeans_index.h:808-829), convergence, reset for the next step
__global__ void k_km_step_end(const float* __restrict__ feat, int n, int C, int K, const double* __restrict__ dc, int* __restrict__ labels, KMState* __restrict__ st, int first, int last) {
    if (st->done) return;
    if (!first) {
        for (int i = 0; i < K; ++i)
            if (st->count[i] == 0) {
                int j = (i + 1) % K, tries = 0;
                while (st->count[j] <= 1 && tries < K) { j = (j + 1) % K; ++tries; }    // bounded: the reference spins forever if no donor exists
                if (st->count[j] <= 1) continue;
                const float rj = __uint_as_float(st->radius[j]);
                for (int k = 0; k < n; ++k)
                    if (labels[k] == j && l2_fd(feat + (size_t)k * C, dc + (size_t)j * C, C) == rj) { labels[k] = i; st->count[j]--; st->count[i]++; break; }
                st->changed = 1;
            }
        if (!st->changed || last) { st->done = 2; return; }
    }
    st->changed = 0;
    for (int i = 0; i < K; ++i) st->radius[i] = 0u;
}
__global__ void k_km_finish(const KMState* __restrict__ st, int K, int* __restrict__ nlabels) { *nlabels = (st->done == 1) ? 1 : K; }

int nctk_kmeans_labels(nct_ctx* ctx, hipStream_t s, const float* feat, int n, int C, int K, int iters, uint64_t seed, int* labels, int* nlabels_dev) {
    NCT_REQUIRE(K >= 1 && K <= KM_MAXK, "kmeans: K=%d out of range", K);
    DevBuf<int> perm(ctx, n);
    DevBuf<double> dc(ctx, (size_t)K * C);
    DevBuf<float> dist(ctx, (size_t)n * K);
    DevBuf<KMState> st(ctx, 1);
    if (!perm.ok() || !dc.ok() || !dist.ok() || !st.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_km_init, dim3(1), dim3(256), 0, s, feat, n, C, K, seed, (int*)perm, (KMState*)st, labels); NCT_LAUNCH_CHECK();
    for (int it = 0; it <= iters; ++it) {          // it == 0: initial assignment to the chosen centres; 1..iters: Lloyd steps
        const int first = it == 0 ? 1 : 0, last = it == iters ? 1 : 0;
        if (!first && C % 256 == 0 && n <= KM_LDS_PERM)
            hipLaunchKernelGGL(k_km_centres_list, dim3(K * C / 256), dim3(256), 0, s, feat, n, C, K, (const int*)labels, (KMState*)st, (double*)dc);
        else
            hipLaunchKernelGGL(k_km_centres, dim3(cdiv(K * C, 256)), dim3(256), 0, s, feat, n, C, K, (const int*)labels, (KMState*)st, (double*)dc, first);
        NCT_LAUNCH_CHECK();
        if (C == KM_DC && KM_DP * K <= 256) {
            const size_t lds = (size_t)K * (KM_DC + KM_CPAD) * sizeof(double) + (size_t)KM_DP * (KM_DC + KM_FPAD) * sizeof(float);      // 41 KB + 50 KB at K = 10
            if (lds > 65536) NCT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km_dist512), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     // per device; a host-side call
            hipLaunchKernelGGL(k_km_dist512, dim3(cdiv(n, KM_DP)), dim3(256), lds, s, feat, n, K, (const double*)dc, (const KMState*)st, (float*)dist);
        } else
        hipLaunchKernelGGL(k_km_dist, dim3(cdiv(n * K, 256)), dim3(256), 0, s, feat, n, C, K, (const double*)dc, (const KMState*)st, (float*)dist);
        NCT_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_km_relabel, dim3(cdiv(n, 256)), dim3(256), 0, s, n, K, (const float*)dist, labels, (KMState*)st, first); NCT_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_km_step_end, dim3(1), dim3(1), 0, s, feat, n, C, K, (const double*)dc, labels, (KMState*)st, first, last); NCT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_km_finish, dim3(1), dim3(1), 0, s, (const KMState*)st, K, nlabels_dev); NCT_LAUNCH_CHECK();
    return 0;
}

// ================================================================= K1: kNN graph
#ifndef NCT_KNN_MAX_BLOCKS
#define NCT_KNN_MAX_BLOCKS 1024
#endif
#ifndef NCT_KNN_LANES16_BELOW
#define NCT_KNN_LANES16_BELOW 31000     // pixels: 44^2 .. 175^2 of a 700^2 pair search with sixteen lanes per entry (k_knn_grid16)
#endif
#ifndef NCT_KNN_RING_UNITS
#define NCT_KNN_RING_UNITS 16
#endif
constexpr int KNN_RING_UNITS = NCT_KNN_RING_UNITS;   // Lab units of Chebyshev rings after which a search scans the rest of its cluster in one pass
constexpr int KNN_K = 8;          // Config.h:68 m_kNum
constexpr int KNN_SLOTS = 5;      // a pixel belongs to its own cluster + at most 4 neighbouring ones

__global__ void k_cell_masks(const int* __restrict__ labels, int lh, int lw, unsigned* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lh * lw) return;
    const int y = i / lw, x = i - y * lw;
    unsigned m = 1u << labels[i];
    if (x < lw - 1) m |= 1u << labels[i + 1];
    if (x > 0) m |= 1u << labels[i - 1];
    if (y < lh - 1) m |= 1u << labels[i + lw];
    if (y > 0) m |= 1u << labels[i - lw];
    mask[i] = m;
}

// ---- (cluster, colour-cell) membership entries: key = cluster << 3*cb | cell(L,a,b), value = pixel id. The cell edge is 2^cs Lab
// units (cb = 8 - cs bits per axis): 32-unit cells (8^3) below 3k pixels, 16-unit below 12k, 8-unit below 100k, 2-unit cells (128^3)
// above: the search is bound by the points it scans in rings 0-1 (4-unit cells: 2.3 ms at 700x700, 8-unit 6.7, 16-unit 23, 2-unit 1.4).
__device__ __forceinline__ unsigned cell_key(int l, unsigned col, int cs) {
    const int cb = 8 - cs;
    return ((unsigned)l << (3 * cb)) | (((col >> 16) & 255u) >> cs) << (2 * cb) | (((col >> 8) & 255u) >> cs) << cb | ((col & 255u) >> cs);
}
// sort key of an entry = cell_key << 3 cs | the colour's position INSIDE its cell (28 bits for every cs): entries of a cell come out ordered by colour, so the 64 queries of
// a wave are (nearly) the same colour — they prune the same neighbouring cells and pass the distance pre-filter together (the search loops are per-thread: what one lane
// cannot skip, the whole wave walks). The order inside a cell does not enter any result.
__device__ __forceinline__ unsigned entry_key(int l, unsigned col, int cs) {
    const unsigned m = (1u << cs) - 1u;
    return (cell_key(l, col, cs) << (3 * cs)) | (((col >> 16) & m) << (2 * cs)) | (((col >> 8) & m) << cs) | (col & m);
}
// entries per pixel = clusters it belongs to; an exclusive scan gives every pixel its place, so the entries are generated in ascending pixel id and the STABLE radix sort
// leaves the ids ascending inside every run of equal (cluster, colour) keys (rounds 1-4 reserved places with one atomic per wave: any order) — the ring search relies on
// that order to take at most k+1 entries of a run (below)
__global__ void k_knn_entry_counts(const unsigned* __restrict__ mask, int lw, int lh, int h, int w, int samples, int nlabels_host, const int* __restrict__ nlabels_dev, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > h * w) return;
    int c = 0;
    if (i < h * w) {
        const int y = i / w, x = i - y * w;
        const int cx = min(x / samples, lw - 1), cy = min(y / samples, lh - 1);
        const int nlabels = nlabels_dev ? *nlabels_dev : nlabels_host;   // the pipeline passes the k-means result without a host round trip
        c = __popc(mask[cy * lw + cx] & ((nlabels >= 32) ? 0xFFFFFFFFu : ((1u << nlabels) - 1u)));
    }
    cnt[i] = c;
}
__global__ void k_knn_entries(const unsigned* __restrict__ mask, const uint8_t* __restrict__ lab, int lw, int lh, int h, int w, int samples, int nlabels_host, const int* __restrict__ nlabels_dev, int cs,
                              const int* __restrict__ off, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const int cx = min(x / samples, lw - 1), cy = min(y / samples, lh - 1);
    const unsigned m = mask[cy * lw + cx];
    const unsigned col = (unsigned)lab[(size_t)i * 3] | ((unsigned)lab[(size_t)i * 3 + 1] << 8) | ((unsigned)lab[(size_t)i * 3 + 2] << 16);
    const int nlabels = nlabels_dev ? *nlabels_dev : nlabels_host;
    int pos = off[i];
    for (int l = 0; l < nlabels; ++l)
        if ((m >> l) & 1u) { keys[pos] = entry_key(l, col, cs); vals[pos] = (unsigned)i; ++pos; }
}
// start[k] = first sorted entry with key >= k, k in [0, nkeys]
// lo/hi (nullable): start table of the 64x coarser cells (keys >> 6) — the search for cell k then stays inside its coarse cell's few entries. With 2-unit cells the
// table has 33.5 M entries, most of them empty cells: a full binary search per cell cost 287 us at 700x700.
__global__ void k_knn_cell_starts(const unsigned* __restrict__ keys, int m, int* __restrict__ start, int nkeys, int shift, const int* __restrict__ coarse) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nkeys) return;
    int lo = 0, hi = m;
    if (coarse) { lo = coarse[k >> 6]; hi = coarse[(k >> 6) + 1]; }
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((keys[mid] >> shift) < (unsigned)k) lo = mid + 1; else hi = mid; }
    start[k] = lo;
}

__device__ __forceinline__ bool ent_less(double d1, int i1, double d2, int i2) { return d1 == d2 ? i1 < i2 : d1 < d2; }
__device__ __forceinline__ double lab_dist(unsigned p, unsigned q) {
    const double s = 1.0 / 255.0;
    const double d0 = (double)(p & 255u) * s - (double)(q & 255u) * s;
    const double d1 = (double)((p >> 8) & 255u) * s - (double)((q >> 8) & 255u) * s;
    const double d2 = (double)((p >> 16) & 255u) * s - (double)((q >> 16) & 255u) * s;
    const double d = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    return d > 0.0 ? d : 0.0;
}

// One thread per (cluster, member) entry. Exact k+1 nearest by (dist, id) inside the cluster: colour cells are visited in
// Chebyshev rings around the query's cell; every point outside ring r differs by at least r*8+1 Lab units in some channel,
// so the search stops as soon as the current (k+1)-th best squared distance is < (r*8+1)^2 (strict: ties cannot hide outside).
// packed Lab colour of every sorted entry: the ring search then streams colours in entry order instead of gathering three bytes per
// scanned point through the pixel id (that gather was ~3/4 of the search time: 14.1 -> 8.7 ms at 700x700)
// + in the top byte: how many entries of the entry's run start at it (capped at 255; ids ascend inside a run): the one-thread search takes at most k+1 of them and jumps
// to the next run — a run of 9 747 identical pixels (in4.png) costs a neighbouring query 39 steps instead of 9 747
__global__ void k_knn_entry_colours(const uint8_t* __restrict__ lab, const int* __restrict__ count, const unsigned* __restrict__ vals, const int* __restrict__ incl,
                                    const int* __restrict__ lead, unsigned* __restrict__ cols) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = *count;
    if (e >= m) return;
    const size_t id = vals[e];
    const int r = incl[e] - 1, nr = incl[m - 1];
    const int rend = r + 1 < nr ? lead[r + 1] : m;
    const int rem = rend - e;
    cols[e] = (unsigned)lab[id * 3] | ((unsigned)lab[id * 3 + 1] << 8) | ((unsigned)lab[id * 3 + 2] << 16) | ((unsigned)(rem < 255 ? rem : 255) << 24);
}
// e = the sorted entry that LEADS a run of equal (cluster, colour) keys, r = the run's index: the k+1 nearest under (dist, id) are a function of the query's cluster and
// colour alone (the query itself is an ordinary candidate; findSubKNNs drops it afterwards), so one search serves the whole run — natural photographs hold runs of 10^4
// pixels (demo/example/in/in4.png: 17 pixels per colour on average, one colour 9 747 times), which made every member scan every other member: 44 ms for one graph.
// The winners are stored as (pixel id, packed colour): k_knn_scatter recomputes the distance with the same lab_dist and hands every member its list minus itself.
// RUNS = false: every entry searches for itself and writes its pixel's candidate slot directly (rounds 1-4; the cheaper form where runs are short: the synthetic pairs
// hold 1.8 pixels per colour, and the run form's extra scatter pass and less coherent waves cost 0.9 ms per pair there). Which form runs is decided ON THE DEVICE from the
// entries-per-run ratio (knn_use_runs): both kernels are launched, the one not chosen returns at once.
struct KnnOut { int* run_id; unsigned* run_col; int* nslot; double* cand_d; int* cand_id; };
__device__ __forceinline__ bool knn_use_runs(int m, int nr, int force) { return force >= 0 ? force != 0 : 2 * (long long)m > 5 * (long long)nr; }   // > 2.5 entries per run
template <bool RUNS>
__device__ __forceinline__ void knn_grid_entry(int e, int r, const unsigned* __restrict__ cols, const unsigned* __restrict__ keys,
                                               const unsigned* __restrict__ vals, const int* __restrict__ start, int cs, const KnnOut& o) {
    const int cb = 8 - cs, CELLS = 1 << cb; const unsigned cmask = (unsigned)CELLS - 1u;
    const unsigned key = keys[e] >> (3 * cs);
    const int id = (int)vals[e];
    const int l = (int)(key >> (3 * cb));
    const int cz = (int)((key >> (2 * cb)) & cmask), cy = (int)((key >> cb) & cmask), cx = (int)(key & cmask);
    const unsigned pc = cols[e] & 0xFFFFFFu;
    double bd[KNN_K + 1]; int bi[KNN_K + 1]; int bq[KNN_K + 1]; unsigned bc[KNN_K + 1];
#pragma unroll
    for (int t = 0; t <= KNN_K; ++t) { bd[t] = 1e300; bi[t] = 0x7fffffff; bq[t] = 0x7fffffff; bc[t] = 0u; }
    // run-wise: the entries of a run share colour and distance and their ids ascend, so the first one that does not enter the list ends the run for this query (an entry
    // that fails on (dist, id) against the current (k+1)-th fails for every larger id as well; the (k+1)-th only improves), and at most k+1 can enter
    // skip_r >= 0 (the fallback pass below): entries whose cell lies inside the cube of rings 0..skip_r were scanned already.
    // Four colour words are requested together (a step that depends on the word just loaded — t += run length — made every entry a full dependent round trip: 1.17 ms
    // instead of 0.97 for HALF the searches at 700x700); an entry that fails takes the rest of its run with it (`adv`), one that enters lets the next one be tested.
    auto scan = [&](int b0, int b1, int skip_r) {
        for (int t = b0; t < b1;) {
            unsigned qw4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) qw4[u] = t + u < b1 ? cols[t + u] : 0u;
            int adv = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t + u >= b1 || u < adv) continue;
                const unsigned qc = qw4[u] & 0xFFFFFFu; const int rem = (int)(qw4[u] >> 24);
                const int e0 = (int)(pc & 255u) - (int)(qc & 255u), e1 = (int)((pc >> 8) & 255u) - (int)((qc >> 8) & 255u), e2 = (int)((pc >> 16) & 255u) - (int)((qc >> 16) & 255u);
                const int q2 = e0 * e0 + e1 * e1 + e2 * e2;
                bool seen = false;
                if (skip_r >= 0) {
                    const int dzc = abs((int)(((qc >> 16) & 255u) >> cs) - cz), dyc = abs((int)(((qc >> 8) & 255u) >> cs) - cy), dxc = abs((int)((qc & 255u) >> cs) - cx);
                    seen = max(dzc, max(dyc, dxc)) <= skip_r;
                }
                bool entered = false;
                if (q2 <= bq[KNN_K] && !seen) {
                    const int jd = (int)vals[t + u];
                    const double d = lab_dist(pc, qc);
                    if (ent_less(d, jd, bd[KNN_K], bi[KNN_K])) {
                        entered = true;
                        bd[KNN_K] = d; bi[KNN_K] = jd; bq[KNN_K] = q2; bc[KNN_K] = qc;
#pragma unroll
                        for (int w = KNN_K; w > 0; --w)
                            if (ent_less(bd[w], bi[w], bd[w - 1], bi[w - 1])) {
                                const double td = bd[w]; bd[w] = bd[w - 1]; bd[w - 1] = td;
                                const int ti = bi[w]; bi[w] = bi[w - 1]; bi[w - 1] = ti;
                                const int tq = bq[w]; bq[w] = bq[w - 1]; bq[w - 1] = tq;
                                const unsigned tc = bc[w]; bc[w] = bc[w - 1]; bc[w - 1] = tc;
                            }
                    }
                }
                // the rest of the run shares this entry's colour and has larger ids: if this one does not enter, none of them does (the (k+1)-th only improves)
                if (!entered) adv = u + rem;
            }
            t += adv > 4 ? adv : 4;
        }
    };
    const int base = l << (3 * cb);
    // Cells that cannot hold a winner are not scanned (round 3): along an axis, a point in cell c differs from the query's coordinate q by at least gap(c) = the distance
    // from q to the cell's interval (0 for the query's own cell), so every point of cell (z, y, x) has d^2 >= gap_z^2 + gap_y^2 + gap_x^2; a cell whose bound EXCEEDS the
    // current (k+1)-th best d^2 can be skipped — its points lose to it even on ties. The bound only tightens while scanning, and the final set is the k+1 smallest under
    // the total order (dist, id) whatever the order of the scan: same ids as the brute-force oracle. With the query's own cell scanned first (a 4-unit cell holds a median of
    // 89 points at 700x700, the 9th-nearest colour is 1-2 units away) most of the 26 cells of ring 1 drop out: ~2400 -> a few hundred points scanned per query.
    const int qz = (int)((pc >> 16) & 255u), qy = (int)((pc >> 8) & 255u), qx = (int)(pc & 255u);   // key order: (col >> 16) is the most significant axis
    auto gap = [&](int q, int c) { const int lo = c << cs, hi = lo + (1 << cs) - 1; return q < lo ? lo - q : (q > hi ? q - hi : 0); };
    // An isolated colour (natural photographs have them; the synthetic pairs' colours are dense) finds its k+1 neighbours only after dozens of rings — (2 r + 1)^2 cell
    // rows each, 45 ms for ONE graph of in4.png while every other lane of the wave waits. After KNN_RING_UNITS Lab units of rings the search therefore falls back to
    // one run-wise pass over the rest of the cluster (entries inside the scanned cube skipped): the same k+1 smallest under (dist, id), bounded work.
    const int rmax = max(2, KNN_RING_UNITS >> cs);
    int rr = 0; bool finished = false;
    for (; rr < CELLS; ++rr) {
        const int r = rr;
        if (r > 0) { const int bound = (r - 1) * (1 << cs) + 1; if (bq[KNN_K] < bound * bound) { finished = true; break; } }
        if (r > rmax) break;
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz; if (z < 0 || z >= CELLS) continue;
            const int gz = gap(qz, z);
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = cy + dy; if (yy < 0 || yy >= CELLS) continue;
                const int gy = gap(qy, yy), gzy = gz * gz + gy * gy;
                if (gzy > bq[KNN_K]) continue;
                const int row = base | (z << (2 * cb)) | (yy << cb);
                if (max(abs(dz), abs(dy)) == r) {                       // full x range of the shell face, minus the cells at its ends that are too far
                    int x0 = max(cx - r, 0), x1 = min(cx + r, CELLS - 1);
                    while (x0 < x1 && gzy + gap(qx, x0) * gap(qx, x0) > bq[KNN_K]) ++x0;
                    while (x1 > x0 && gzy + gap(qx, x1) * gap(qx, x1) > bq[KNN_K]) --x1;
                    if (gzy + gap(qx, x0) * gap(qx, x0) > bq[KNN_K]) continue;              // x0 == x1 and that cell is too far as well
                    scan(start[row | x0], start[(row | x1) + 1], -1);
                } else {                                                 // only the two end cells
                    if (cx - r >= 0 && gzy + gap(qx, cx - r) * gap(qx, cx - r) <= bq[KNN_K]) scan(start[row | (cx - r)], start[(row | (cx - r)) + 1], -1);
                    if (cx + r < CELLS && gzy + gap(qx, cx + r) * gap(qx, cx + r) <= bq[KNN_K]) scan(start[row | (cx + r)], start[(row | (cx + r)) + 1], -1);
                }
            }
        }
    }
    if (!finished && rr < CELLS) scan(start[base], start[base + (1 << (3 * cb))], rr - 1);      // rings 0 .. rr - 1 are done
    if constexpr (RUNS) {
#pragma unroll
        for (int t = 0; t <= KNN_K; ++t) { o.run_id[(size_t)r * (KNN_K + 1) + t] = bi[t]; o.run_col[(size_t)r * (KNN_K + 1) + t] = bc[t]; }
    } else {
        const int slot = atomicAdd(&o.nslot[id], 1);
        double* od = o.cand_d + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
        int* oi = o.cand_id + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
        int ni = 0;                                                  // findSubKNNs: drop self, keep the first k
#pragma unroll
        for (int t = 0; t <= KNN_K; ++t)
            if (bi[t] != id && bi[t] != 0x7fffffff && ni < KNN_K) { od[ni] = bd[t]; oi[ni] = bi[t]; ++ni; }
        for (; ni < KNN_K; ++ni) { od[ni] = 1e300; oi[ni] = -1; }
    }
}
// The same search by SIXTEEN lanes per entry, for the coarse levels (a few thousand queries: one thread per entry leaves the chip to a few hundred waves that each
// walk hundreds of points one after the other — 430 us for the 1936 pixels of 44 x 44, and the first nonlocal solve waits for exactly that graph). Lane v scans the
// points whose index in the sorted array is v modulo 16 into its OWN (k+1)-list (ownership by absolute index: lanes prune on different thresholds, so they do not scan the same ranges); a list only ever holds points that were the best seen by its lane, so its (k+1)-th entry
// bounds the true one from above and every pruning decision a lane takes on it is valid; at the end of a ring the sixteen lists are merged (nine rounds of a 16-lane
// butterfly minimum under (dist, id); entries the lanes share since the last merge pop together) and every lane continues from the merged list. The result is the set
// of the k+1 smallest under the total order: the same ids as the one-thread form and the brute-force oracle.
template <bool RUNS>
__device__ __forceinline__ void knn_grid_entry16(int e, int r, int v, const unsigned* __restrict__ cols, const unsigned* __restrict__ keys,
                                                 const unsigned* __restrict__ vals, const int* __restrict__ start, int cs, const KnnOut& o) {
    const int cb = 8 - cs, CELLS = 1 << cb; const unsigned cmask = (unsigned)CELLS - 1u;
    const unsigned key = keys[e] >> (3 * cs);
    const int id = (int)vals[e];
    const int l = (int)(key >> (3 * cb));
    const int cz = (int)((key >> (2 * cb)) & cmask), cy = (int)((key >> cb) & cmask), cx = (int)(key & cmask);
    const unsigned pc = cols[e] & 0xFFFFFFu;
    double bd[KNN_K + 1]; int bi[KNN_K + 1]; int bq[KNN_K + 1]; unsigned bc[KNN_K + 1];
#pragma unroll
    for (int t = 0; t <= KNN_K; ++t) { bd[t] = 1e300; bi[t] = 0x7fffffff; bq[t] = 0x7fffffff; bc[t] = 0u; }
    auto scan = [&](int b0, int b1, int skip_r) {
        for (int t = b0 + ((v - b0) & 15); t < b1; t += 16) {     // lane v owns the points with index = v (mod 16), whatever range ITS thresholds make it scan
            const unsigned qc = cols[t] & 0xFFFFFFu;
            const int e0 = (int)(pc & 255u) - (int)(qc & 255u), e1 = (int)((pc >> 8) & 255u) - (int)((qc >> 8) & 255u), e2 = (int)((pc >> 16) & 255u) - (int)((qc >> 16) & 255u);
            const int q2 = e0 * e0 + e1 * e1 + e2 * e2;
            if (q2 > bq[KNN_K]) continue;
            if (skip_r >= 0) {
                const int dzc = abs((int)(((qc >> 16) & 255u) >> cs) - cz), dyc = abs((int)(((qc >> 8) & 255u) >> cs) - cy), dxc = abs((int)((qc & 255u) >> cs) - cx);
                if (max(dzc, max(dyc, dxc)) <= skip_r) continue;
            }
            const int jd = (int)vals[t];
            const double d = lab_dist(pc, qc);
            if (!ent_less(d, jd, bd[KNN_K], bi[KNN_K])) continue;
            bd[KNN_K] = d; bi[KNN_K] = jd; bq[KNN_K] = q2; bc[KNN_K] = qc;
#pragma unroll
            for (int u = KNN_K; u > 0; --u)
                if (ent_less(bd[u], bi[u], bd[u - 1], bi[u - 1])) {
                    const double td = bd[u]; bd[u] = bd[u - 1]; bd[u - 1] = td;
                    const int ti = bi[u]; bi[u] = bi[u - 1]; bi[u - 1] = ti;
                    const int tq = bq[u]; bq[u] = bq[u - 1]; bq[u - 1] = tq;
                    const unsigned tc = bc[u]; bc[u] = bc[u - 1]; bc[u - 1] = tc;
                }
        }
    };
    auto merge16 = [&]() {
        double od[KNN_K + 1]; int oi[KNN_K + 1], oq[KNN_K + 1]; unsigned oc[KNN_K + 1];
#pragma unroll
        for (int s = 0; s <= KNN_K; ++s) {
            double md = bd[0]; int mi = bi[0], mq = bq[0]; unsigned mc = bc[0];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const double xd = __shfl_xor(md, off, 16); const int xi = __shfl_xor(mi, off, 16), xq = __shfl_xor(mq, off, 16); const unsigned xc = __shfl_xor(mc, off, 16);
                if (ent_less(xd, xi, md, mi)) { md = xd; mi = xi; mq = xq; mc = xc; }
            }
            od[s] = md; oi[s] = mi; oq[s] = mq; oc[s] = mc;
            if (bi[0] == mi) {                                        // this lane's head is the winner (or every list is exhausted): pop
#pragma unroll
                for (int u = 0; u < KNN_K; ++u) { bd[u] = bd[u + 1]; bi[u] = bi[u + 1]; bq[u] = bq[u + 1]; bc[u] = bc[u + 1]; }
                bd[KNN_K] = 1e300; bi[KNN_K] = 0x7fffffff; bq[KNN_K] = 0x7fffffff; bc[KNN_K] = 0u;
            }
        }
#pragma unroll
        for (int s = 0; s <= KNN_K; ++s) { bd[s] = od[s]; bi[s] = oi[s]; bq[s] = oq[s]; bc[s] = oc[s]; }
    };
    const int base = l << (3 * cb);
    const int qz = (int)((pc >> 16) & 255u), qy = (int)((pc >> 8) & 255u), qx = (int)(pc & 255u);
    auto gap = [&](int q, int c) { const int lo = c << cs, hi = lo + (1 << cs) - 1; return q < lo ? lo - q : (q > hi ? q - hi : 0); };
    const int rmax = max(2, KNN_RING_UNITS >> cs);         // then one pass over the rest of the cluster (see knn_grid_entry)
    int rr = 0; bool finished = false;
    for (; rr < CELLS; ++rr) {
        const int r = rr;
        if (r > 0) { const int bound = (r - 1) * (1 << cs) + 1; if (bq[KNN_K] < bound * bound) { finished = true; break; } }   // merged list: the same value in all sixteen lanes
        if (r > rmax) break;
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz; if (z < 0 || z >= CELLS) continue;
            const int gz = gap(qz, z);
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = cy + dy; if (yy < 0 || yy >= CELLS) continue;
                const int gy = gap(qy, yy), gzy = gz * gz + gy * gy;
                if (gzy > bq[KNN_K]) continue;
                const int row = base | (z << (2 * cb)) | (yy << cb);
                if (max(abs(dz), abs(dy)) == r) {
                    int x0 = max(cx - r, 0), x1 = min(cx + r, CELLS - 1);
                    while (x0 < x1 && gzy + gap(qx, x0) * gap(qx, x0) > bq[KNN_K]) ++x0;
                    while (x1 > x0 && gzy + gap(qx, x1) * gap(qx, x1) > bq[KNN_K]) --x1;
                    if (gzy + gap(qx, x0) * gap(qx, x0) > bq[KNN_K]) continue;
                    scan(start[row | x0], start[(row | x1) + 1], -1);
                } else {
                    if (cx - r >= 0 && gzy + gap(qx, cx - r) * gap(qx, cx - r) <= bq[KNN_K]) scan(start[row | (cx - r)], start[(row | (cx - r)) + 1], -1);
                    if (cx + r < CELLS && gzy + gap(qx, cx + r) * gap(qx, cx + r) <= bq[KNN_K]) scan(start[row | (cx + r)], start[(row | (cx + r)) + 1], -1);
                }
            }
        }
        merge16();
    }
    if (!finished && rr < CELLS) { scan(start[base], start[base + (1 << (3 * cb))], rr - 1); merge16(); }
    if (v != 0) return;
    if constexpr (RUNS) {
#pragma unroll
        for (int t = 0; t <= KNN_K; ++t) { o.run_id[(size_t)r * (KNN_K + 1) + t] = bi[t]; o.run_col[(size_t)r * (KNN_K + 1) + t] = bc[t]; }
    } else {
        const int slot = atomicAdd(&o.nslot[id], 1);
        double* od = o.cand_d + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
        int* oi = o.cand_id + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
        int ni = 0;
#pragma unroll
        for (int t = 0; t <= KNN_K; ++t)
            if (bi[t] != id && bi[t] != 0x7fffffff && ni < KNN_K) { od[ni] = bd[t]; oi[ni] = bi[t]; ++ni; }
        for (; ni < KNN_K; ++ni) { od[ni] = 1e300; oi[ni] = -1; }
    }
}
// runs of equal sorted keys = equal (cluster, colour): flag of the leading entry; an inclusive scan of the flags numbers the runs in entry order (the leaders keep
// the cell order that lets a wave's 64 searches walk the same cells — an unordered compaction lost exactly that)
__global__ void k_knn_run_flags(const unsigned* __restrict__ keys, const int* __restrict__ count, int cap, int* __restrict__ flag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cap) return;
    flag[e] = (e < *count && (e == 0 || keys[e] != keys[e - 1])) ? 1 : 0;
}
__global__ void k_knn_run_leads(const int* __restrict__ flag, const int* __restrict__ incl, const int* __restrict__ count, int* __restrict__ lead) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= *count) return;
    if (flag[e]) lead[incl[e] - 1] = e;
}
template <bool RUNS>
__global__ __launch_bounds__(256) void k_knn_grid16(const unsigned* __restrict__ cols, const int* __restrict__ count, const int* __restrict__ incl, const int* __restrict__ lead,
                                                    const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, const int* __restrict__ start, int cs, int force, KnnOut o) {
    const int m = *count; const int nr = m > 0 ? incl[m - 1] : 0;
    if (knn_use_runs(m, nr, force) != RUNS) return;
    const int v = threadIdx.x & 15;
    const int nq = RUNS ? nr : m;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < nq; r += gridDim.x * 16)
        knn_grid_entry16<RUNS>(RUNS ? lead[r] : r, r, v, cols, keys, vals, start, cs, o);
}
// Grid-stride over the runs (entries) with a BOUNDED grid: the searches are long-running and this kernel lives on the side stream; a grid
// that fills every CU slot makes the short main-stream kernels wait until all of its workgroups have been dispatched.
template <bool RUNS>
__global__ __launch_bounds__(256) void k_knn_grid(const unsigned* __restrict__ cols, const int* __restrict__ count, const int* __restrict__ incl, const int* __restrict__ lead,
                                                  const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, const int* __restrict__ start, int cs, int force, KnnOut o) {
    const int m = *count; const int nr = m > 0 ? incl[m - 1] : 0;
    if (knn_use_runs(m, nr, force) != RUNS) return;
    const int nq = RUNS ? nr : m;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < nq; r += gridDim.x * 256)
        knn_grid_entry<RUNS>(RUNS ? lead[r] : r, r, cols, keys, vals, start, cs, o);
}
// every entry takes its run's k+1 winners minus itself (findSubKNNs: drop self, keep the first k) into one of its pixel's candidate slots
__global__ void k_knn_scatter(const int* __restrict__ count, const int* __restrict__ incl, const unsigned* __restrict__ cols, const unsigned* __restrict__ vals, int force,
                              const int* __restrict__ run_id, const unsigned* __restrict__ run_col, int* __restrict__ nslot, double* __restrict__ cand_d, int* __restrict__ cand_id) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = *count;
    if (e >= m || !knn_use_runs(m, incl[m - 1], force)) return;
    const int r = incl[e] - 1;
    const int id = (int)vals[e];
    const unsigned pc = cols[e] & 0xFFFFFFu;
    const int slot = atomicAdd(&nslot[id], 1);
    double* od = cand_d + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
    int* oi = cand_id + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
    int ni = 0;
#pragma unroll
    for (int t = 0; t <= KNN_K; ++t) {
        const int bi = run_id[(size_t)r * (KNN_K + 1) + t];
        if (bi != id && bi != 0x7fffffff && ni < KNN_K) { od[ni] = lab_dist(pc, run_col[(size_t)r * (KNN_K + 1) + t]); oi[ni] = bi; ++ni; }
    }
    for (; ni < KNN_K; ++ni) { od[ni] = 1e300; oi[ni] = -1; }
}

// sortMergeComputeWeight: sort by (dist,id), dedupe, keep k, w = exp(1 - d/3); pad with zero-weight self edges
__global__ void k_knn_merge(int npix, const int* __restrict__ nslot, const double* __restrict__ cand_d, const int* __restrict__ cand_id,
                            int* __restrict__ knn_id, double* __restrict__ knn_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    // every slot list is already sorted by (dist, id) with -1 padding at the end: a <=5-way merge with the heads in registers
    // emits the unique entries in ascending (dist, id) order (= sort + dedupe of the reference; equal ids carry equal distances)
    const int ns = nslot[i];
    const double* cd = cand_d + (size_t)i * KNN_SLOTS * KNN_K;
    const int* ci = cand_id + (size_t)i * KNN_SLOTS * KNN_K;
    double hd[KNN_SLOTS]; int hi[KNN_SLOTS], pos[KNN_SLOTS];
#pragma unroll
    for (int sl = 0; sl < KNN_SLOTS; ++sl) {
        pos[sl] = 0; hi[sl] = -1; hd[sl] = 1e300;
        if (sl < ns) { hi[sl] = ci[sl * KNN_K]; hd[sl] = cd[sl * KNN_K]; }
    }
    int lp = 0;
    while (lp < KNN_K) {
        double bd = 1e300; int bid = -1;
#pragma unroll
        for (int sl = 0; sl < KNN_SLOTS; ++sl)
            if (hi[sl] >= 0 && (bid < 0 || ent_less(hd[sl], hi[sl], bd, bid))) { bd = hd[sl]; bid = hi[sl]; }
        if (bid < 0) break;
        knn_id[(size_t)i * KNN_K + lp] = bid;
        knn_w[(size_t)i * KNN_K + lp] = nct_exp(1.0 - bd / 3.0);
        ++lp;
#pragma unroll
        for (int sl = 0; sl < KNN_SLOTS; ++sl)
            if (hi[sl] == bid) {
                ++pos[sl];
                if (pos[sl] < KNN_K) { hi[sl] = ci[sl * KNN_K + pos[sl]]; hd[sl] = cd[sl * KNN_K + pos[sl]]; } else hi[sl] = -1;
            }
    }
    for (; lp < KNN_K; ++lp) { knn_id[(size_t)i * KNN_K + lp] = i; knn_w[(size_t)i * KNN_K + lp] = 0.0; }
}

int nctk_knn_graph(nct_ctx* ctx, hipStream_t s, const uint8_t* lab_u8, int h, int w, const int* labels, int lh, int lw, int nlabels, const int* nlabels_dev, int samples,
                   int* knn_id, double* knn_w) {
    NCT_REQUIRE(nlabels_dev || (nlabels >= 1 && nlabels <= 16), "knn_graph: nlabels=%d out of range", nlabels);
    const int n = h * w;
    // cell edge 2^cs Lab units, chosen so that a cell holds a handful of points: sparse (coarse-level) point sets in fine cells make a
    // query walk thousands of empty cells before it has seen k+1 points
    const int cs = n >= 100000 ? 1 : n >= 12000 ? 3 : n >= 3000 ? 4 : 5, cb = 8 - cs;
    const int cap = n * KNN_SLOTS, nkeys = 16 << (3 * cb);
    const unsigned key_sentinel = 1u << 28;                 // sorts after every real key (16 clusters x cells x in-cell position = 4 + 3 cb + 3 cs = 28 bits)
    DevBuf<unsigned> mask(ctx, (size_t)lh * lw), keys(ctx, cap), vals(ctx, cap), keys_s(ctx, cap), vals_s(ctx, cap);
    DevBuf<unsigned> cols(ctx, cap);
    DevBuf<int> start(ctx, nkeys + 2), cstart(ctx, (nkeys >> 6) + 4), nslot(ctx, n), cand_id(ctx, (size_t)n * KNN_SLOTS * KNN_K);
    DevBuf<double> cand_d(ctx, (size_t)n * KNN_SLOTS * KNN_K);
    if (!mask.ok() || !keys.ok() || !vals.ok() || !keys_s.ok() || !vals_s.ok() || !cols.ok() || !start.ok() || !cstart.ok() || !nslot.ok() || !cand_id.ok() || !cand_d.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemsetAsync(nslot, 0, sizeof(int) * n, s));
    NCT_HIP(hipMemsetD32Async((hipDeviceptr_t)(unsigned*)keys, (int)key_sentinel, cap, s));       // unused slots sort to the end
    hipLaunchKernelGGL(k_cell_masks, dim3(cdiv(lh * lw, 256)), dim3(256), 0, s, labels, lh, lw, (unsigned*)mask);
    NCT_LAUNCH_CHECK();
    DevBuf<int> ecnt(ctx, (size_t)n + 1), eoff(ctx, (size_t)n + 1);
    if (!ecnt.ok() || !eoff.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_knn_entry_counts, dim3(cdiv(n + 1, 256)), dim3(256), 0, s, (const unsigned*)mask, lw, lh, h, w, samples, nlabels, nlabels_dev, (int*)ecnt); NCT_LAUNCH_CHECK();
    {
        size_t sb = 0;
        NCT_HIP(rocprim::exclusive_scan(nullptr, sb, (const int*)ecnt, (int*)eoff, 0, (size_t)n + 1, rocprim::plus<int>(), s));
        DevBuf<char> st(ctx, sb + 16);
        if (!st.ok()) return NCT_ERR_HIP;
        NCT_HIP(rocprim::exclusive_scan((void*)(char*)st, sb, (const int*)ecnt, (int*)eoff, 0, (size_t)n + 1, rocprim::plus<int>(), s));
    }
    const int* count = (const int*)eoff + n;                // number of entries
    hipLaunchKernelGGL(k_knn_entries, dim3(cdiv(n, 256)), dim3(256), 0, s, (const unsigned*)mask, lab_u8, lw, lh, h, w, samples, nlabels, nlabels_dev, cs,
                       (const int*)eoff, (unsigned*)keys, (unsigned*)vals);
    NCT_LAUNCH_CHECK();
    size_t tmp_bytes = 0;
    NCT_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned*)keys, (unsigned*)keys_s, (const unsigned*)vals, (unsigned*)vals_s, cap, 0, 29, s));
    DevBuf<char> tmp(ctx, tmp_bytes ? tmp_bytes : 16);
    if (!tmp.ok()) return NCT_ERR_HIP;
    NCT_HIP(rocprim::radix_sort_pairs((void*)(char*)tmp, tmp_bytes, (const unsigned*)keys, (unsigned*)keys_s, (const unsigned*)vals, (unsigned*)vals_s, cap, 0, 29, s));
    if (nkeys >= (1 << 22)) {                            // two stages: 64x coarser cells first
        hipLaunchKernelGGL(k_knn_cell_starts, dim3(cdiv((nkeys >> 6) + 2, 256)), dim3(256), 0, s, (const unsigned*)keys_s, cap, (int*)cstart, (nkeys >> 6) + 1, 3 * cs + 6, (const int*)nullptr);
        NCT_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_knn_cell_starts, dim3(cdiv(nkeys + 1, 256)), dim3(256), 0, s, (const unsigned*)keys_s, cap, (int*)start, nkeys, 3 * cs, (const int*)cstart);
    } else
        hipLaunchKernelGGL(k_knn_cell_starts, dim3(cdiv(nkeys + 1, 256)), dim3(256), 0, s, (const unsigned*)keys_s, cap, (int*)start, nkeys, 3 * cs, (const int*)nullptr);
    NCT_LAUNCH_CHECK();
    // one search per run of equal (cluster, colour) entries
    DevBuf<int> flag(ctx, cap), incl(ctx, cap), lead(ctx, cap), run_id(ctx, (size_t)cap * (KNN_K + 1));
    DevBuf<unsigned> run_col(ctx, (size_t)cap * (KNN_K + 1));
    if (!flag.ok() || !incl.ok() || !lead.ok() || !run_id.ok() || !run_col.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_knn_run_flags, dim3(cdiv(cap, 256)), dim3(256), 0, s, (const unsigned*)keys_s, count, cap, (int*)flag); NCT_LAUNCH_CHECK();
    size_t scan_bytes = 0;
    NCT_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, (const int*)flag, (int*)incl, (size_t)cap, rocprim::plus<int>(), s));
    DevBuf<char> tmp2(ctx, scan_bytes + 16);
    if (!tmp2.ok()) return NCT_ERR_HIP;
    NCT_HIP(rocprim::inclusive_scan((void*)(char*)tmp2, scan_bytes, (const int*)flag, (int*)incl, (size_t)cap, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_knn_run_leads, dim3(cdiv(cap, 256)), dim3(256), 0, s, (const int*)flag, (const int*)incl, count, (int*)lead); NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_knn_entry_colours, dim3(cdiv(cap, 256)), dim3(256), 0, s, lab_u8, count, (const unsigned*)vals_s, (const int*)incl, (const int*)lead, (unsigned*)cols);
    NCT_LAUNCH_CHECK();
    const KnnOut ko{run_id, run_col, nslot, cand_d, cand_id};
    const int force = ctx->knn_runs;                        // -1: by the entries-per-run ratio, on the device; 0 / 1: NCT_KNN_RUNS (tests)
#define NCT_KNN_LAUNCH(RUNS) do { \
    if (n <= NCT_KNN_LANES16_BELOW) \
        hipLaunchKernelGGL(k_knn_grid16<RUNS>, dim3(std::min(cdiv(cap, 16), 4 * NCT_KNN_MAX_BLOCKS)), dim3(256), 0, s, (const unsigned*)cols, count, (const int*)incl, (const int*)lead, \
                           (const unsigned*)keys_s, (const unsigned*)vals_s, (const int*)start, cs, force, ko); \
    else \
        hipLaunchKernelGGL(k_knn_grid<RUNS>, dim3(std::min(cdiv(cap, 256), NCT_KNN_MAX_BLOCKS)), dim3(256), 0, s, (const unsigned*)cols, count, (const int*)incl, (const int*)lead, \
                           (const unsigned*)keys_s, (const unsigned*)vals_s, (const int*)start, cs, force, ko); \
    NCT_LAUNCH_CHECK(); } while (0)
    NCT_KNN_LAUNCH(false);
    NCT_KNN_LAUNCH(true);
#undef NCT_KNN_LAUNCH
    hipLaunchKernelGGL(k_knn_scatter, dim3(cdiv(cap, 256)), dim3(256), 0, s, count, (const int*)incl, (const unsigned*)cols, (const unsigned*)vals_s, force,
                       (const int*)run_id, (const unsigned*)run_col, (int*)nslot, (double*)cand_d, (int*)cand_id);
    NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_knn_merge, dim3(cdiv(n, 256)), dim3(256), 0, s, n, (const int*)nslot, (const double*)cand_d, (const int*)cand_id, knn_id, knn_w);
    NCT_LAUNCH_CHECK();
    return 0;
}
