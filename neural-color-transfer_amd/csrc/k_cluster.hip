// k_cluster.hip — C1 (k-means labels of the coarsest S features) and K1 (per-cluster kNN graph in Lab).
// Reference: ColorTransfer::clusterFeastures ColorTransfer.cpp:355-395 -> cvflann KMeansIndex::computeClustering
// (Flann/kmeans_index.h:700-880, chooseCentersRandom :108-135, L2 distance Flann/dist.h:153-181);
// ColorTransfer::findKnns/getClusters/findSubKNNs/sortMergeComputeWeight ColorTransfer.cpp:397-423,273-353,136-195,60-110
// (nanoflann KD-trees, one per cluster, built and queried on the host under OpenMP in the reference).
//
// MI355X design:
//  * k-means: the problem is tiny (<= 63x63 points x 512-d, k=10, <= 11 Lloyd steps) and strictly sequential between
//    steps, so it runs as ONE persistent 1024-thread workgroup: no host round trips, fp64 centre sums accumulated in
//    point order exactly like the reference loop (bit-identical labels to the oracle).
//  * kNN: exact brute force per cluster instead of KD-trees. Lab comes from 8-bit images, so the squared distance is an
//    exact small integer: candidates are streamed through LDS as packed u32 (L,a,b), pre-filtered with integer
//    arithmetic against the current 9th-best, and only survivors pay for the fp64 sqrt that defines the final
//    (dist, id) order (= the reference's cmpDist order; ties are first-come in the reference's KD-tree).
#include "nct_internal.h"
#include "nct_device.h"
#include "nct_detmath.h"

// ================================================================= C1: k-means
__device__ __forceinline__ uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ float l2_fd(const float* __restrict__ a, const double* __restrict__ b, int n) {
    float result = 0.f;
    int i = 0;
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
    for (; i + 3 < n; i += 4) {
        const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; i < n; ++i) { const float d0 = a[i] - b[i]; result += d0 * d0; }
    return result;
}

constexpr int KM_MAXK = 16;
__global__ __launch_bounds__(1024) void k_kmeans(const float* __restrict__ feat, int n, int C, int K, int iters, uint64_t seed,
                                                 int* __restrict__ labels, int* __restrict__ nlabels, int* __restrict__ perm, double* __restrict__ dc) {
    __shared__ int s_cidx[KM_MAXK], s_count[KM_MAXK], s_flag, s_nc;
    __shared__ unsigned s_radius[KM_MAXK];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n; i += nt) labels[i] = 0;
    if (tid == 0) {
        s_nc = 0;
        if (n >= K) {
            for (int i = 0; i < n; ++i) perm[i] = i;
            uint64_t st = seed;
            for (int i = n - 1; i > 0; --i) { const int j = (int)(sm64(st) % (uint64_t)(i + 1)); const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
            int pos = 0, nc = 0; bool out = false;
            for (int index = 0; index < K && !out; ++index) {
                bool dup = true;
                while (dup) {
                    dup = false;
                    if (pos >= n) { out = true; break; }
                    s_cidx[index] = perm[pos++];
                    for (int j = 0; j < index; ++j)
                        if (l2_ff(feat + (size_t)s_cidx[index] * C, feat + (size_t)s_cidx[j] * C, C) < 1e-16) dup = true;
                }
                if (!out) nc = index + 1;
            }
            s_nc = nc;
        }
    }
    __syncthreads();
    if (s_nc < K) { if (tid == 0) *nlabels = 1; return; }
    for (int i = tid; i < K * C; i += nt) { const int c = i / C, k = i - c * C; dc[i] = (double)feat[(size_t)s_cidx[c] * C + k]; }
    if (tid < K) { s_count[tid] = 0; s_radius[tid] = 0u; }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const float* v = feat + (size_t)i * C;
        float sq = l2_fd(v, dc, C); int b = 0;
        for (int j = 1; j < K; ++j) { const float nsq = l2_fd(v, dc + (size_t)j * C, C); if (sq > nsq) { b = j; sq = nsq; } }
        labels[i] = b;
        atomicMax(&s_radius[b], __float_as_uint(sq));
        atomicAdd(&s_count[b], 1);
    }
    __syncthreads();
    for (int iteration = 0; iteration < iters; ++iteration) {
        if (tid == 0) s_flag = 1;               // converged
        // new centres: fp64 sums in point order, one (cluster, dim) pair per thread slice
        for (int i = tid; i < K * C; i += nt) {
            const int c = i / C, k = i - c * C;
            double s = 0.0;
            for (int p = 0; p < n; ++p) if (labels[p] == c) s += (double)feat[(size_t)p * C + k];
            dc[i] = s / (double)s_count[c];
        }
        if (tid < K) s_radius[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            const float* v = feat + (size_t)i * C;
            float sq = l2_fd(v, dc, C); int b = 0;
            for (int j = 1; j < K; ++j) { const float nsq = l2_fd(v, dc + (size_t)j * C, C); if (sq > nsq) { b = j; sq = nsq; } }
            atomicMax(&s_radius[b], __float_as_uint(sq));
            const int old = labels[i];
            if (b != old) { atomicSub(&s_count[old], 1); atomicAdd(&s_count[b], 1); labels[i] = b; s_flag = 0; }
        }
        __syncthreads();
        if (tid == 0) {      // an emptied cluster takes the farthest point of the next cluster with > 1 members
            for (int i = 0; i < K; ++i)
                if (s_count[i] == 0) {
                    int j = (i + 1) % K;
                    while (s_count[j] <= 1) j = (j + 1) % K;
                    const float rj = __uint_as_float(s_radius[j]);
                    for (int k = 0; k < n; ++k)
                        if (labels[k] == j && l2_fd(feat + (size_t)k * C, dc + (size_t)j * C, C) == rj) { labels[k] = i; s_count[j]--; s_count[i]++; break; }
                    s_flag = 0;
                }
        }
        __syncthreads();
        const int conv = s_flag;
        __syncthreads();
        if (conv) break;
    }
    if (tid == 0) *nlabels = K;
}

int nctk_kmeans_labels(nct_ctx* ctx, hipStream_t s, const float* feat, int n, int C, int K, int iters, uint64_t seed, int* labels, int* nlabels_dev) {
    NCT_REQUIRE(K >= 1 && K <= KM_MAXK, "kmeans: K=%d out of range", K);
    DevBuf<int> perm(ctx, n);
    DevBuf<double> dc(ctx, (size_t)K * C);
    if (!perm.ok() || !dc.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_kmeans, dim3(1), dim3(1024), 0, s, feat, n, C, K, iters, seed, labels, nlabels_dev, (int*)perm, (double*)dc);
    NCT_LAUNCH_CHECK();
    return 0;
}

// ================================================================= K1: kNN graph
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

__global__ void k_cluster_members(const unsigned* __restrict__ mask, int lw, int lh, int h, int w, int samples, int nlabels,
                                  int* __restrict__ cnt, int* __restrict__ mem) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const int cx = min(x / samples, lw - 1), cy = min(y / samples, lh - 1);
    unsigned m = mask[cy * lw + cx];
    for (int l = 0; l < nlabels; ++l)
        if ((m >> l) & 1u) { const int pos = atomicAdd(&cnt[l], 1); mem[(size_t)l * h * w + pos] = i; }
}

struct KnnEnt { double d; int id; };
__device__ __forceinline__ bool ent_less(double d1, int i1, double d2, int i2) { return d1 == d2 ? i1 < i2 : d1 < d2; }
__device__ __forceinline__ double lab_dist(unsigned p, unsigned q) {
    const double s = 1.0 / 255.0;
    const double d0 = (double)(p & 255u) * s - (double)(q & 255u) * s;
    const double d1 = (double)((p >> 8) & 255u) * s - (double)((q >> 8) & 255u) * s;
    const double d2 = (double)((p >> 16) & 255u) * s - (double)((q >> 16) & 255u) * s;
    const double d = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    return d > 0.0 ? d : 0.0;
}

// one thread per member query; the cluster's members stream through LDS in tiles of 256
__global__ __launch_bounds__(256) void k_knn_cluster(const uint8_t* __restrict__ lab, int npix, const int* __restrict__ cnt, const int* __restrict__ mem,
                                                     int* __restrict__ nslot, double* __restrict__ cand_d, int* __restrict__ cand_id) {
    const int l = blockIdx.y;
    const int m = cnt[l];
    const int q0 = blockIdx.x * 256;
    if (q0 >= m) return;
    const int* members = mem + (size_t)l * npix;
    __shared__ unsigned s_col[256];
    __shared__ int s_id[256];
    const int qi = q0 + threadIdx.x;
    const bool live = qi < m;
    const int id = live ? members[qi] : -1;
    unsigned pc = 0;
    if (live) pc = (unsigned)lab[(size_t)id * 3] | ((unsigned)lab[(size_t)id * 3 + 1] << 8) | ((unsigned)lab[(size_t)id * 3 + 2] << 16);
    double bd[KNN_K + 1]; int bi[KNN_K + 1]; int bq[KNN_K + 1];      // distance, id, integer squared distance
#pragma unroll
    for (int t = 0; t <= KNN_K; ++t) { bd[t] = 1e300; bi[t] = 0x7fffffff; bq[t] = 0x7fffffff; }
    for (int t0 = 0; t0 < m; t0 += 256) {
        __syncthreads();
        const int j = t0 + threadIdx.x;
        if (j < m) {
            const int jd = members[j];
            s_id[threadIdx.x] = jd;
            s_col[threadIdx.x] = (unsigned)lab[(size_t)jd * 3] | ((unsigned)lab[(size_t)jd * 3 + 1] << 8) | ((unsigned)lab[(size_t)jd * 3 + 2] << 16);
        }
        __syncthreads();
        if (!live) continue;
        const int tn = min(256, m - t0);
        for (int t = 0; t < tn; ++t) {
            const unsigned qc = s_col[t];
            const int e0 = (int)(pc & 255u) - (int)(qc & 255u), e1 = (int)((pc >> 8) & 255u) - (int)((qc >> 8) & 255u), e2 = (int)((pc >> 16) & 255u) - (int)((qc >> 16) & 255u);
            const int q2 = e0 * e0 + e1 * e1 + e2 * e2;
            if (q2 > bq[KNN_K]) continue;                       // exact integer pre-filter
            const int jd = s_id[t];
            const double d = lab_dist(pc, qc);
            if (!ent_less(d, jd, bd[KNN_K], bi[KNN_K])) continue;
            bd[KNN_K] = d; bi[KNN_K] = jd; bq[KNN_K] = q2;
#pragma unroll
            for (int u = KNN_K; u > 0; --u)
                if (ent_less(bd[u], bi[u], bd[u - 1], bi[u - 1])) {
                    const double td = bd[u]; bd[u] = bd[u - 1]; bd[u - 1] = td;
                    const int ti = bi[u]; bi[u] = bi[u - 1]; bi[u - 1] = ti;
                    const int tq = bq[u]; bq[u] = bq[u - 1]; bq[u - 1] = tq;
                }
        }
    }
    if (!live) return;
    const int slot = atomicAdd(&nslot[id], 1);
    double* od = cand_d + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
    int* oi = cand_id + ((size_t)id * KNN_SLOTS + slot) * KNN_K;
    int ni = 0;                                                  // findSubKNNs: drop self, keep the first k
#pragma unroll
    for (int t = 0; t <= KNN_K; ++t)
        if (bi[t] != id && bi[t] != 0x7fffffff && ni < KNN_K) { od[ni] = bd[t]; oi[ni] = bi[t]; ++ni; }
    for (; ni < KNN_K; ++ni) { od[ni] = 1e300; oi[ni] = -1; }
}

// sortMergeComputeWeight: sort by (dist,id), dedupe, keep k, w = exp(1 - d/3); pad with zero-weight self edges
__global__ void k_knn_merge(int npix, const int* __restrict__ nslot, const double* __restrict__ cand_d, const int* __restrict__ cand_id,
                            int* __restrict__ knn_id, double* __restrict__ knn_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int total = nslot[i] * KNN_K;
    const double* cd = cand_d + (size_t)i * KNN_SLOTS * KNN_K;
    const int* ci = cand_id + (size_t)i * KNN_SLOTS * KNN_K;
    double last_d = -1.0; int last_id = -1, lp = 0;
    // selection by repeated minimum over <= 40 entries strictly greater than the last emitted (dist,id)
    while (lp < KNN_K) {
        double bd = 1e300; int bid = -1;
        for (int t = 0; t < total; ++t) {
            const int id = ci[t];
            if (id < 0) continue;
            const double d = cd[t];
            const bool after = (last_id < 0) || ent_less(last_d, last_id, d, id);
            if (after && (bid < 0 || ent_less(d, id, bd, bid))) { bd = d; bid = id; }
        }
        if (bid < 0) break;
        // the reference dedupes on consecutive equal ids after sorting by (dist,id): equal ids have equal distances
        knn_id[(size_t)i * KNN_K + lp] = bid;
        knn_w[(size_t)i * KNN_K + lp] = nct_exp(1.0 - bd / 3.0);
        last_d = bd; last_id = bid; ++lp;
    }
    for (; lp < KNN_K; ++lp) { knn_id[(size_t)i * KNN_K + lp] = i; knn_w[(size_t)i * KNN_K + lp] = 0.0; }
}

int nctk_knn_graph(nct_ctx* ctx, hipStream_t s, const uint8_t* lab_u8, int h, int w, const int* labels, int lh, int lw, int nlabels, int samples,
                   int* knn_id, double* knn_w) {
    NCT_REQUIRE(nlabels >= 1 && nlabels <= 16, "knn_graph: nlabels=%d out of range", nlabels);
    const int n = h * w;
    DevBuf<unsigned> mask(ctx, (size_t)lh * lw);
    DevBuf<int> cnt(ctx, 16), mem(ctx, (size_t)nlabels * n), nslot(ctx, n), cand_id(ctx, (size_t)n * KNN_SLOTS * KNN_K);
    DevBuf<double> cand_d(ctx, (size_t)n * KNN_SLOTS * KNN_K);
    if (!mask.ok() || !cnt.ok() || !mem.ok() || !nslot.ok() || !cand_id.ok() || !cand_d.ok()) return NCT_ERR_HIP;
    NCT_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * 16, s));
    NCT_HIP(hipMemsetAsync(nslot, 0, sizeof(int) * n, s));
    hipLaunchKernelGGL(k_cell_masks, dim3(cdiv(lh * lw, 256)), dim3(256), 0, s, labels, lh, lw, (unsigned*)mask);
    NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cluster_members, dim3(cdiv(n, 256)), dim3(256), 0, s, (const unsigned*)mask, lw, lh, h, w, samples, nlabels, (int*)cnt, (int*)mem);
    NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_knn_cluster, dim3(cdiv(n, 256), nlabels), dim3(256), 0, s, lab_u8, n, (const int*)cnt, (const int*)mem, (int*)nslot,
                       (double*)cand_d, (int*)cand_id);
    NCT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_knn_merge, dim3(cdiv(n, 256)), dim3(256), 0, s, n, (const int*)nslot, (const double*)cand_d, (const int*)cand_id, knn_id, knn_w);
    NCT_LAUNCH_CHECK();
    return 0;
}
