// k_s1.hip — S1: the nonlocal least squares of the local colour model, solved by the reference's truncated, un-preconditioned CG.
// Reference: solve_nonlocal_downsample_gpu_gradient ColorTransfer.cpp:548-949 (assembly: data rows :611-658, local smoothness rows :660-847, kNN rows :849-911,
//            iteration caps :916-921) -> solve_ls_cg_gpu SparseSolver_GPU.cu:3-198 (explicit A^T A by SpGEMM, CG loop :132-159).
//
// MI355X design
//  * matrix free: A has <= 2 non-zeros per row, so A^T A is (a) a 2x2 data block per pixel and channel, (b) twice the 5-point graph Laplacian with weights
//    g^2 and (c) the kNN graph Laplacian. Out-edges come from the [n][8] kNN table, in-edges from a radix-sorted reverse adjacency. The three Lab channels run
//    in lock step with their own CG scalars; it is the SAME truncated recurrence started from the local-stats guess and stopped by the 50 / 100 iteration cap.
//  * SINGLE-REDUCTION recurrence (round 5; Chronopoulos & Gear): one operator pass computes w = Op(r) and BOTH dot products (gamma = r.r, delta = w.r), one
//    vector pass updates p = r + beta p, s = w + beta s (= Op(p)), x += alpha p, r -= alpha s. Two launches per iteration (three at the finest level, whose 1915
//    block partials are reduced by a one-workgroup kernel) instead of three (five), the vectors are streamed once per iteration instead of twice. Same iterates
//    in exact arithmetic; the rounding order is this project's own specification (oracle/orc_color_canon.c mirrors it operation for operation).
//  * HUBS (round 5). In natural photographs thousands of pixels share one colour; the kNN tie rule (distance, id) then makes the lowest ids of such a group the
//    neighbours of the whole group: in-degrees of 10^4 (demo/example/in/in1.png: 33 335 at 700x528; the synthetic pairs: <= 37). One thread walking such a list made
//    the operator 100x slower (1.06 s instead of 8 ms for the finest level). The in-edge sum is therefore specified in blocks of S1_SEG = 64 edges: the first block
//    is added edge by edge inside the operator pass as before (lists of <= 64 entries are summed exactly as in rounds 1-4), every further block is summed by a
//    64-leaf tree — one wave per block, one gather per lane, k_s1_hub — in a pass of its own in front of the operator pass, whose pixel thread then adds the block
//    sums in order (pixels with MORE than 64 further blocks — in-degree above 4 160: a letterboxed frame, a flat background — add the sums of SUPER-blocks of 64 block sums
//    instead, k_s1_hub2: the same tree one level up). Whether a level has such blocks is known to the HOST without a synchronisation: the graph-only part of the system (reverse adjacency, block
//    table) is built on the side stream right behind the level's kNN graph and publishes the block count into page-locked memory; by the time the host enqueues a
//    level's solve the count has long arrived (checked with hipEventQuery; if not, the hub pass is launched anyway — it exits on the device-side count).
//    Hub-free levels (every level of the synthetic pairs) launch nothing extra.
// Roofline: the operator is a gather kernel (16 random 48-byte records per pixel out of a 23.5 MB vector at 700x700: fabric bound), the vector
// pass a pure stream (432 B per pixel).
#include "nct_internal.h"
#include "nct_device.h"
#include "nct_reduce.h"
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>   // rocPRIM directly (no CUB-compatibility layer)
#include <rocprim/device/device_scan.hpp>

#define LAB_D(u) ((double)(u) * (1.0 / 255.0))      // Mat::convertTo(CV_64F, 1/255)
#define LCHK() NCT_LAUNCH_CHECK()

constexpr int S1_SEG = NCT_S1_SEG;                   // nct_internal.h: 64

struct S1Sys {
    int n, h, w;
    const double *daa, *dab, *dbb;          // [n][3]: (dw s)^2, (dw s) dw, dw^2
    const double *gx, *gy;                  // [n]
    const int* knn_id;                      // [n][8]
    nct_s1_graph g;                         // iw2, compact in-edge arrays, hub block table
    int xcd;                                // workgroups of one XCD take a contiguous range of pixel blocks (the shared-gather levels; NCT_S1_XCD=0: plain order)
    int one_xcd;                            // small levels: 1 + h = the launch is 8 x the grid and only the workgroups that land on XCD h work (s1_one_xcd below); 0: off
};
// Small levels (<= 32 workgroups: 44 x 44 and 88 x 88 of a 700 x 700 pair) are pure latency chains — own record -> neighbour ids -> gathers -> in-edge ids -> gathers —
// and what the chain fetches was written by the PREVIOUS launch (the vector pass writes r, the operator pass w). Workgroup b runs on XCD b % 8 and the eight L2s are not
// coherent with each other: a line written on another XCD comes from memory (Infinity Cache), a line written on the same XCD from its L2 (MI355X_MICROARCH.md: same-XCD
// hand-offs 1.7 x faster). So both passes are launched with 8 x the workgroups, those with b % 8 != 0 exit at once, and the solve's vectors (1.9 MB at 88 x 88) live in
// ONE L2 for the whole solve (the context's home XCD: contexts of a process take different ones, so four pairs in flight do not queue on the same 32 CUs). Placement is an
// observed property, not a promise: it changes the time, never the result. Measured (700 x 700 pair, one in flight): 3.17 -> 2.03 ms at 44 x 44, 2.04 -> 1.74 at 88 x 88;
// 175 x 175 (120 workgroups, 7 MB of vectors) does not fit one XCD: 2.2 -> 5.2 ms. sel = 1 + home XCD. -1: not a working workgroup.
__device__ __forceinline__ int s1_one_xcd(int bid, int sel) { return (bid & 7) == sel - 1 ? (bid >> 3) : -1; }
// workgroup ids go round-robin over the 8 XCDs: XCD x takes the x-th eighth of the logical blocks
__device__ __forceinline__ int s1_block_of(int bid, int nblocks, int xcd) {
    if (!xcd) return bid;
    const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, idx = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
}
struct S1State { double gm[3], al[3], be[3]; int active[3], iters[3]; };

// a [pixel][6] record as three 16-byte loads / stores (records are 48 bytes, arena blocks 256-byte aligned)
__device__ __forceinline__ void ld6(const double* __restrict__ v, size_t px, double (&o)[6]) {
    const double2* q = reinterpret_cast<const double2*>(v + px * 6);
    const double2 t0 = q[0], t1 = q[1], t2 = q[2];
    o[0] = t0.x; o[1] = t0.y; o[2] = t1.x; o[3] = t1.y; o[4] = t2.x; o[5] = t2.y;
}
__device__ __forceinline__ void st6(double* __restrict__ v, size_t px, const double (&o)[6]) {
    double2* q = reinterpret_cast<double2*>(v + px * 6);
    q[0] = make_double2(o[0], o[1]); q[1] = make_double2(o[2], o[3]); q[2] = make_double2(o[4], o[5]);
}
__device__ __forceinline__ int lo32(unsigned long long v) { return (int)(unsigned)(v & 0xFFFFFFFFull); }
__device__ __forceinline__ int hi32(unsigned long long v) { return (int)(unsigned)(v >> 32); }

// ---------------------------------------------------------------- graph-only part (side stream in the pipeline)
__global__ void k_s1_iw2(const double* __restrict__ knn_w, int m, double nonlocalWeight, double* __restrict__ iw2) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const double iw = sqrt(knn_w[e]) * nonlocalWeight;
    iw2[e] = iw * iw;
}
__global__ void k_edge_keys(const int* __restrict__ knn_id, int m, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    keys[e] = (unsigned)knn_id[e]; vals[e] = (unsigned)e;
}
__global__ void k_seg_starts(const unsigned* __restrict__ keys, int m, int* __restrict__ start, int n) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n) return;
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < (unsigned)s) lo = mid + 1; else hi = mid; }
    start[s] = lo;
}
// per pixel: (entries of its first block, number of further blocks) packed lo / hi; one exclusive scan turns both into start offsets
__global__ void k_s1_counts(const int* __restrict__ rev_start, int n, unsigned long long* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    unsigned long long v = 0;
    if (i < n) {
        const int deg = rev_start[i + 1] - rev_start[i];
        v = (unsigned long long)(deg < S1_SEG ? deg : S1_SEG) | ((unsigned long long)(deg > S1_SEG ? (deg - 1) / S1_SEG : 0) << 32);
    }
    cnt[i] = v;
}
// pixels with more than 64 hub blocks: super-blocks of 64 blocks (one exclusive scan of ceil(blocks / 64)); thread per pixel fills its super-blocks' first block indices
__global__ void k_s1_sup_counts(const unsigned long long* __restrict__ starts, int n, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    int c = 0;
    if (i < n) { const int m = hi32(starts[i + 1]) - hi32(starts[i]); c = m > S1_SEG ? (m + S1_SEG - 1) / S1_SEG : 0; }
    cnt[i] = c;
}
__global__ void k_s1_sup_fill(const unsigned long long* __restrict__ starts, const int* __restrict__ sup_start, int n, int* __restrict__ sup_b0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s0 = sup_start[i], ns = sup_start[i + 1] - s0, b0 = hi32(starts[i]);
    for (int k = 0; k < ns; ++k) sup_b0[s0 + k] = b0 + k * S1_SEG;
}
// position e of the target-sorted edge list (ascending edge id src*8+ki inside a target): rank r inside its target's list. r < 64 -> the compact arrays the
// operator pass reads; r >= 64 -> the full-position arrays the hub pass reads, and the first edge of every further block fills the block table
__global__ void k_s1_rev_build(const unsigned* __restrict__ keys_sorted, const unsigned* __restrict__ edge_sorted, const double* __restrict__ iw2, int m,
                               const int* __restrict__ rev_start, const unsigned long long* __restrict__ starts,
                               int* __restrict__ c_src, double* __restrict__ c_w, int* __restrict__ rev_src, double* __restrict__ rev_w,
                               int* __restrict__ seg_tgt, int* __restrict__ seg_e0) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const int t = (int)keys_sorted[e];
    const int rank = e - rev_start[t];
    const unsigned ed = edge_sorted[e];
    const int src = (int)(ed >> 3); const double wt = iw2[ed];
    const unsigned long long st = starts[t];
    if (rank < S1_SEG) { const int pos = lo32(st) + rank; c_src[pos] = src; c_w[pos] = wt; }
    else {
        rev_src[e] = src; rev_w[e] = wt;
        if ((rank & (S1_SEG - 1)) == 0) { const int sg = hi32(st) + rank / S1_SEG - 1; seg_tgt[sg] = t; seg_e0[sg] = e; }
    }
}

int nctk_s1_graph_build(nct_ctx* ctx, hipStream_t s, const int* knn_id, const double* knn_w, double nonlocalWeight, const nct_s1_graph& g, int* nseg_pinned) {
    const int n = g.n, m = 8 * n;
    DevBuf<unsigned> ek(ctx, m), ev(ctx, m), eks(ctx, m), evs(ctx, m);
    DevBuf<unsigned long long> cnt(ctx, (size_t)n + 1);
    DevBuf<int> scnt(ctx, (size_t)n + 1);
    if (!ek.ok() || !ev.ok() || !eks.ok() || !evs.ok() || !cnt.ok() || !scnt.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_s1_iw2, dim3(cdiv(m, 256)), dim3(256), 0, s, knn_w, m, nonlocalWeight, g.iw2); LCHK();
    hipLaunchKernelGGL(k_edge_keys, dim3(cdiv(m, 256)), dim3(256), 0, s, knn_id, m, (unsigned*)ek, (unsigned*)ev); LCHK();
    int end_bit = 1; while ((1u << end_bit) < (unsigned)n && end_bit < 32) ++end_bit;
    size_t tmp_bytes = 0, scan_bytes = 0, scan2_bytes = 0;
    NCT_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned*)ek, (unsigned*)eks, (const unsigned*)ev, (unsigned*)evs, m, 0, end_bit, s));
    NCT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (const unsigned long long*)cnt, g.starts, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    NCT_HIP(rocprim::exclusive_scan(nullptr, scan2_bytes, (const int*)scnt, g.sup_start, 0, (size_t)n + 1, rocprim::plus<int>(), s));
    if (scan2_bytes > scan_bytes) scan_bytes = scan2_bytes;
    DevBuf<char> tmp(ctx, (tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes) + 16);
    if (!tmp.ok()) return NCT_ERR_HIP;
    NCT_HIP(rocprim::radix_sort_pairs((void*)(char*)tmp, tmp_bytes, (const unsigned*)ek, (unsigned*)eks, (const unsigned*)ev, (unsigned*)evs, m, 0, end_bit, s));
    hipLaunchKernelGGL(k_seg_starts, dim3(cdiv(n + 1, 256)), dim3(256), 0, s, (const unsigned*)eks, m, g.rev_start, n); LCHK();
    hipLaunchKernelGGL(k_s1_counts, dim3(cdiv(n + 1, 256)), dim3(256), 0, s, (const int*)g.rev_start, n, (unsigned long long*)cnt); LCHK();
    NCT_HIP(rocprim::exclusive_scan((void*)(char*)tmp, scan_bytes, (const unsigned long long*)cnt, g.starts, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    hipLaunchKernelGGL(k_s1_rev_build, dim3(cdiv(m, 256)), dim3(256), 0, s, (const unsigned*)eks, (const unsigned*)evs, (const double*)g.iw2, m, (const int*)g.rev_start,
                       (const unsigned long long*)g.starts, g.c_src, g.c_w, g.rev_src, g.rev_w, g.seg_tgt, g.seg_e0); LCHK();
    hipLaunchKernelGGL(k_s1_sup_counts, dim3(cdiv(n + 1, 256)), dim3(256), 0, s, (const unsigned long long*)g.starts, n, (int*)scnt); LCHK();
    NCT_HIP(rocprim::exclusive_scan((void*)(char*)tmp, scan2_bytes, (const int*)scnt, g.sup_start, 0, (size_t)n + 1, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_s1_sup_fill, dim3(cdiv(n, 256)), dim3(256), 0, s, (const unsigned long long*)g.starts, (const int*)g.sup_start, n, g.sup_b0); LCHK();
    // the numbers of hub blocks (high half of the last scan element) and super-blocks travel to the host behind this work; the caller reads them only after an
    // event recorded behind this call has completed
    if (nseg_pinned) {
        NCT_HIP(hipMemcpyAsync(nseg_pinned, (const char*)(g.starts + n) + 4, sizeof(int), hipMemcpyDeviceToHost, s));
        NCT_HIP(hipMemcpyAsync(nseg_pinned + 1, g.sup_start + n, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    return 0;
}

// ---------------------------------------------------------------- data part of the system
__global__ void k_s1_setup(int n, const double* __restrict__ weight, float dWeight, const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref,
                           double* __restrict__ daa, double* __restrict__ dab, double* __restrict__ dbb, double* __restrict__ rhs /*[2][n][3]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dw = sqrt(weight[i]) * (double)sqrtf(dWeight);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v0 = dw * LAB_D(src[(size_t)i * 3 + c]);
        const double rb = dw * LAB_D(ref[(size_t)i * 3 + c]);
        daa[(size_t)i * 3 + c] = v0 * v0; dab[(size_t)i * 3 + c] = v0 * dw; dbb[(size_t)i * 3 + c] = dw * dw;
        rhs[(size_t)i * 3 + c] = v0 * rb; rhs[(size_t)(n + i) * 3 + c] = dw * rb;
    }
}

// ---------------------------------------------------------------- hub pass: one wave per block of 64 in-edges beyond a pixel's first block
// lane j: term of edge j of the block (or +0 behind the list's end), then the 64-leaf halving tree as an xor butterfly (lane 0 ends with exactly the tree's
// s[0]: at every step it adds the partner's value of the step before, and a + b == b + a)
__global__ __launch_bounds__(256) void k_s1_hub(nct_s1_graph G, const double* __restrict__ v /*[n][6]*/) {
    const int nseg = hi32(G.starts[G.n]);
    const int lane = threadIdx.x & 63;
    for (int sg = blockIdx.x * 4 + (threadIdx.x >> 6); sg < nseg; sg += gridDim.x * 4) {
        const int t = G.seg_tgt[sg];
        const int e = G.seg_e0[sg] + lane;
        const bool in = e < G.rev_start[t + 1];
        double term[6], vt[6], vj[6];
        int j = t; double wt = 0.0;
        if (in) { j = G.rev_src[e]; wt = G.rev_w[e]; }
        ld6(v, (size_t)t, vt); ld6(v, (size_t)j, vj);
#pragma unroll
        for (int c = 0; c < 6; ++c) { const double d = wt * (vt[c] - vj[c]); term[c] = in ? d : 0.0; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int c = 0; c < 6; ++c) term[c] += __shfl_xor(term[c], off);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) G.hub_part[(size_t)sg * 6 + c] = term[c];
        }
    }
}

// second level: one wave per super-block of 64 block sums (contiguous records of hub_part), the same 64-leaf tree. Only pixels with more than 64 hub blocks — in-degree
// above 4160: a letterboxed frame's black bars, a flat background — have super-blocks; their pixel thread then adds <= a few dozen super-block sums instead of thousands of
// block sums one after the other (700x700 with 30 % black bars: 2 039 blocks on each of nine hubs, the operator pass 488 instead of 147 us)
__global__ __launch_bounds__(256) void k_s1_hub2(nct_s1_graph G) {
    const int nsup = G.sup_start[G.n];
    const int lane = threadIdx.x & 63;
    for (int sb = blockIdx.x * 4 + (threadIdx.x >> 6); sb < nsup; sb += gridDim.x * 4) {
        // the super-block's pixel is not stored: its last block is bounded by the next super-block's first (same pixel) or, for a pixel's last super-block, by that
        // pixel's block range — found through the block table's target
        const int b0 = G.sup_b0[sb];
        const int t = G.seg_tgt[b0];
        const int bend = min(b0 + S1_SEG, hi32(G.starts[t + 1]));
        double term[6];
        const bool in = b0 + lane < bend;
        if (in) ld6(G.hub_part, (size_t)(b0 + lane), term);
#pragma unroll
        for (int c = 0; c < 6; ++c) term[c] = in ? term[c] : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int c = 0; c < 6; ++c) term[c] += __shfl_xor(term[c], off);
        if (lane == 0) st6(G.sup_part, (size_t)sb, term);
    }
}

// ---------------------------------------------------------------- operator
// y = Op(v) at pixel i (live = i < n); must be called by every thread of a 256-thread workgroup whose threads own consecutive pixels. a, b return the pixel's own
// record. Order per pixel (= oracle/orc_color_canon.c s1_op): data block; +x, -x, +y, -y raster neighbours with weight 2 g^2 (every edge is entered twice in A,
// ColorTransfer.cpp:671-843); the 8 kNN out-edges; the first <= 64 in-edges in ascending edge id; the sums of the further blocks of 64 in block order (k_s1_hub).
// The in-degree is mild on the synthetic pairs (mean 8, p99 19, max 37) but an in-edge is a dependent random 48-byte gather, so at the bandwidth-bound levels
// (COOP) the gathers are shared: the first-block in-edges of a workgroup's 256 consecutive pixels are ONE contiguous range of the compact arrays; the threads
// fetch it edge-parallel into LDS in chunks of S1_CHUNK edges, then every thread adds ITS edges from LDS in edge order — same per-pixel order, same bits.
#ifndef NCT_S1_CHUNK
#define NCT_S1_CHUNK 1024
#endif
constexpr int S1_CHUNK = NCT_S1_CHUNK;
template <bool COOP>
__device__ __forceinline__ void s1_op(const S1Sys& S, const double* __restrict__ p, int i, bool live, double (&a)[3], double (&b)[3], double (&ya)[3], double (&yb)[3], int lb = -1) {
    const int w = S.w, h = S.h;
    int e0 = 0, e1 = 0, h0 = 0, h1 = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] = 0.0; b[c] = 0.0; ya[c] = 0.0; yb[c] = 0.0; }
    if (live) {
        const int y = i / w, x = i - y * w;
        // the gathered vector is interleaved [pixel][a0 a1 a2 b0 b1 b2]: one 48-byte read per neighbour instead of two 24-byte ones
        { double own[6]; ld6(p, (size_t)i, own);
#pragma unroll
          for (int c = 0; c < 3; ++c) { a[c] = own[c]; b[c] = own[3 + c]; } }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ya[c] = S.daa[(size_t)i * 3 + c] * a[c] + S.dab[(size_t)i * 3 + c] * b[c];
            yb[c] = S.dab[(size_t)i * 3 + c] * a[c] + S.dbb[(size_t)i * 3 + c] * b[c];
        }
        auto edge = [&](int j, double wt) {
            double q[6]; ld6(p, (size_t)j, q);
#pragma unroll
            for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - q[c]); yb[c] += wt * (b[c] - q[3 + c]); }
        };
        // raster neighbours: the four gradient weights and the four records are requested TOGETHER (an absent neighbour reads the pixel itself and is not added) — written as
        // `if (exists) { g = gx[..]; edge(..) }` each term was a branch of its own with two dependent round trips inside: eight serial trips before the first kNN gather
        // (round 6, from the ISA; 0.85 of the 3.15 ms of the 44 x 44 level). Added in the same order (+x, -x, +y, -y): same bits.
        {
            const bool ex[4] = {x + 1 < w, x > 0, y + 1 < h, y > 0};
            const int nj[4] = {ex[0] ? i + 1 : i, ex[1] ? i - 1 : i, ex[2] ? i + w : i, ex[3] ? i - w : i};
            const double g4[4] = {S.gx[i], S.gx[ex[1] ? i - 1 : i], S.gy[i], S.gy[ex[3] ? i - w : i]};
            double q4[4][6];
#pragma unroll
            for (int u = 0; u < 4; ++u) ld6(p, (size_t)nj[u], q4[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ex[u]) {
                    const double wt = 2.0 * (g4[u] * g4[u]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - q4[u][c]); yb[c] += wt * (b[c] - q4[u][3 + c]); }
                }
        }
        // nonlocal: out-edges (8 independent gathers per thread), then in-edges
#pragma unroll
        for (int k = 0; k < 8; ++k) edge(S.knn_id[(size_t)i * 8 + k], S.g.iw2[(size_t)i * 8 + k]);
        const unsigned long long s0 = S.g.starts[i], s1 = S.g.starts[i + 1];
        e0 = lo32(s0); e1 = lo32(s1); h0 = hi32(s0); h1 = hi32(s1);
    }
    if constexpr (!COOP) {
        // small levels (latency bound, few workgroups): every thread walks its own list, loads of four edges issued together
        if (live) {
            int e = e0;
            for (; e + 4 <= e1; e += 4) {
                int j[4]; double wt[4], pv[4][6];
#pragma unroll
                for (int u = 0; u < 4; ++u) { j[u] = S.g.c_src[e + u]; wt[u] = S.g.c_w[e + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) ld6(p, (size_t)j[u], pv[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ya[c] += wt[u] * (a[c] - pv[u][c]); yb[c] += wt[u] * (b[c] - pv[u][3 + c]); }
            }
            for (; e < e1; ++e) {
                const int j = S.g.c_src[e]; const double wt = S.g.c_w[e];
                double q[6]; ld6(p, (size_t)j, q);
#pragma unroll
                for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - q[c]); yb[c] += wt * (b[c] - q[3 + c]); }
            }
        }
    } else {
        // first-block in-edges of the workgroup's pixels [i0, i1): compact range [E0, E1)
        __shared__ double s_pv[6 * S1_CHUNK];          // [c][edge]
        __shared__ double s_wt[S1_CHUNK];
        const int i0 = (lb >= 0 ? lb : (int)blockIdx.x) * 256, i1 = min(i0 + 256, S.n);
        const int E0 = lo32(S.g.starts[i0]), E1 = lo32(S.g.starts[i1]);
        for (int base = E0; base < E1; base += S1_CHUNK) {
            const int cnt = min(S1_CHUNK, E1 - base);
            for (int t = threadIdx.x; t < cnt; t += 256) {
                const int j = S.g.c_src[base + t];
                s_wt[t] = S.g.c_w[base + t];
                double q[6]; ld6(p, (size_t)j, q);
#pragma unroll
                for (int c = 0; c < 6; ++c) s_pv[c * S1_CHUNK + t] = q[c];
            }
            __syncthreads();
            const int lo = max(e0, base) - base, hi = min(e1, base + cnt) - base;
            for (int t = lo; t < hi; ++t) {
                const double wt = s_wt[t];
#pragma unroll
                for (int c = 0; c < 3; ++c) { ya[c] += wt * (a[c] - s_pv[c * S1_CHUNK + t]); yb[c] += wt * (b[c] - s_pv[(3 + c) * S1_CHUNK + t]); }
            }
            __syncthreads();
        }
    }
    // hub pixels: the sums of their further blocks of 64 in-edges, in block order (h0 == h1 for every pixel of a hub-free level); with more than 64 blocks, the sums of
    // their super-blocks of 64 blocks instead (k_s1_hub2)
    bool sup = false;
    if (h1 - h0 > S1_SEG) { h0 = S.g.sup_start[i]; h1 = S.g.sup_start[i + 1]; sup = true; }
    const double* __restrict__ part = sup ? S.g.sup_part : S.g.hub_part;
    for (int sg = h0; sg < h1; ++sg) {
        double q[6]; ld6(part, (size_t)sg, q);
#pragma unroll
        for (int c = 0; c < 3; ++c) { ya[c] += q[c]; yb[c] += q[3 + c]; }
    }
}

// r = rhs - Op(x0)
template <bool COOP>
__global__ __launch_bounds__(256) void k_s1_residual(S1Sys S, const double* __restrict__ x6, const double* __restrict__ rhs, double* __restrict__ r6) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double a[3], b[3], ya[3], yb[3];
    s1_op<COOP>(S, x6, i, i < S.n, a, b, ya, yb);
    if (i < S.n) {
        double o[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = rhs[(size_t)i * 3 + c] - ya[c]; o[3 + c] = rhs[(size_t)(S.n + i) * 3 + c] - yb[c]; }
        st6(r6, (size_t)i, o);
    }
}
// w = Op(r); block partials of gamma = r.r (slots 0..2) and delta = r.w (slots 3..5), per Lab channel over both parts
template <bool COOP>
__global__ __launch_bounds__(256) void k_s1_apply(S1Sys S, const double* __restrict__ r6, double* __restrict__ w6, double* __restrict__ partial /*[nb][6]*/) {
    const int lb = S.one_xcd ? s1_one_xcd(blockIdx.x, S.one_xcd) : s1_block_of(blockIdx.x, gridDim.x, S.xcd);
    if (lb < 0) return;
    const int i = lb * 256 + threadIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    double a[3], b[3], ya[3], yb[3];
    s1_op<COOP>(S, r6, i, i < S.n, a, b, ya, yb, lb);
    if (i < S.n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[c] = a[c] * a[c] + b[c] * b[c];
            acc[3 + c] = a[c] * ya[c] + b[c] * yb[c];
        }
        const double o[6] = {ya[0], ya[1], ya[2], yb[0], yb[1], yb[2]};
        st6(w6, (size_t)i, o);
    }
    block_reduce_store<6>(acc, partial, lb);
}

// ---------------------------------------------------------------- scalars and vector pass
// the state after operator pass j from its two sums and the state after pass j - 1 (first: j == 0). iters counts the vector passes the state will drive.
__device__ __forceinline__ void s1_scalars(bool first, const double (&sm)[6], const S1State* __restrict__ in, double tol2, S1State& o) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (first) {
            const double g0 = sm[c]; const bool act = g0 > tol2;
            o.gm[c] = g0; o.be[c] = 0.0; o.al[c] = act ? g0 / sm[3 + c] : 0.0; o.active[c] = act ? 1 : 0; o.iters[c] = act ? 1 : 0;
        } else if (in->active[c]) {
            const double g1 = sm[c];
            const double be = g1 / in->gm[c];
            const double al = g1 / (sm[3 + c] - (be * g1) / in->al[c]);
            const bool act = g1 > tol2;
            o.gm[c] = g1; o.be[c] = be; o.al[c] = al; o.active[c] = act ? 1 : 0; o.iters[c] = in->iters[c] + (act ? 1 : 0);
        } else {
            o.gm[c] = in->gm[c]; o.be[c] = in->be[c]; o.al[c] = in->al[c]; o.active[c] = 0; o.iters[c] = in->iters[c];
        }
    }
}
__global__ __launch_bounds__(256) void k_s1_scal(const double* __restrict__ partial, int nb, const S1State* __restrict__ in, S1State* __restrict__ out, int first, double tol2) {
    double sm[6]; final_reduce<6>(partial, nb, sm);
    if (threadIdx.x == 0) { S1State o; s1_scalars(first != 0, sm, in, tol2, o); *out = o; }
}
// vector pass, thread per pixel, every vector interleaved [pixel][6]: p = r + beta p, s = w + beta s, x += alpha p, r -= alpha s (first_vec: p = r, s = w).
// FUSED (levels with <= S1_FUSE_NB block partials): every workgroup repeats the fixed-order final reduction of the operator pass's partials (a few KB out of L2)
// and derives the scalars itself — the one-workgroup kernel between the two passes disappears; the state is double buffered (workgroup 0 writes `sout` while
// the others still read `sin`). The pixel's operands do not depend on the scalars: they are requested in front of the reduction and fly under it.
constexpr int S1_FUSE_NB = 512;
// Every vector is [pixel][6] and the update is elementwise with per-channel scalars (channel of flat element f = f % 3, because 6 = 0 mod 3): the kernel walks the
// vectors as flat arrays of double2 — lane-consecutive 16-byte accesses — three per thread (a workgroup = the 768 double2 of 256 pixels). A thread-per-pixel form with
// 8-byte accesses at a 48-byte lane stride ran at 2.5 TB/s (84 us at 700x700); this one is a plain stream.
template <bool FUSED>
__global__ __launch_bounds__(256) void k_s1_update(int n, int nb, const double* __restrict__ partial, const S1State* __restrict__ sin, S1State* __restrict__ sout,
                                                   int first_scal, double tol2, int first_vec, double* __restrict__ r6, const double* __restrict__ w6,
                                                   double* __restrict__ p6, double* __restrict__ s6, double* __restrict__ x6, int one_xcd) {
    const int bid = one_xcd ? s1_one_xcd(blockIdx.x, one_xcd) : (int)blockIdx.x;
    if (bid < 0) return;
    const size_t total = (size_t)3 * n;                       // double2 elements per vector
    const size_t base = (size_t)bid * 768 + threadIdx.x;
    double2 rv[3], wv[3], pv[3], sv[3], xv[3];
    const double2* r2 = reinterpret_cast<const double2*>(r6); const double2* w2 = reinterpret_cast<const double2*>(w6);
    const double2* x2 = reinterpret_cast<const double2*>(x6); const double2* p2 = reinterpret_cast<const double2*>(p6); const double2* s2 = reinterpret_cast<const double2*>(s6);
    // (all fifteen loads in one round trip: elements beyond the end — the last workgroup only — read the vector's last element and are not written. As `if (g < total) { loads }`
    //  per k the three groups were three dependent round trips: round 6, from the ISA)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t g0 = base + (size_t)k * 256, g = g0 < total ? g0 : total - 1;
        rv[k] = r2[g]; wv[k] = w2[g]; xv[k] = x2[g];
        pv[k] = p2[g]; sv[k] = s2[g];
        if (first_vec) { pv[k] = make_double2(0.0, 0.0); sv[k] = make_double2(0.0, 0.0); }
    }
    double al[3], be[3]; bool act[3];
    if constexpr (FUSED) {
        double sm[6]; final_reduce<6>(partial, nb, sm);
        S1State o; s1_scalars(first_scal != 0, sm, sin, tol2, o);
        if (bid == 0 && threadIdx.x == 0) *sout = o;
#pragma unroll
        for (int c = 0; c < 3; ++c) { al[c] = o.al[c]; be[c] = o.be[c]; act[c] = o.active[c] != 0; }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { al[c] = sin->al[c]; be[c] = sin->be[c]; act[c] = sin->active[c] != 0; }
    }
    double2* r2o = reinterpret_cast<double2*>(r6); double2* x2o = reinterpret_cast<double2*>(x6); double2* p2o = reinterpret_cast<double2*>(p6); double2* s2o = reinterpret_cast<double2*>(s6);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t g = base + (size_t)k * 256;
        if (g >= total) continue;
        const int c0 = (int)((2 * g) % 3), c1 = (c0 + 1) % 3;
        // per element exactly the oracle's expressions; an inactive channel keeps its four values
        const double a0 = c0 == 0 ? al[0] : (c0 == 1 ? al[1] : al[2]), b0 = c0 == 0 ? be[0] : (c0 == 1 ? be[1] : be[2]);
        const double a1 = c1 == 0 ? al[0] : (c1 == 1 ? al[1] : al[2]), b1 = c1 == 0 ? be[0] : (c1 == 1 ? be[1] : be[2]);
        const bool on0 = c0 == 0 ? act[0] : (c0 == 1 ? act[1] : act[2]), on1 = c1 == 0 ? act[0] : (c1 == 1 ? act[1] : act[2]);
        double2 pn, sn, xn, rn;
        pn.x = first_vec ? rv[k].x : b0 * pv[k].x + rv[k].x;  pn.y = first_vec ? rv[k].y : b1 * pv[k].y + rv[k].y;
        sn.x = first_vec ? wv[k].x : b0 * sv[k].x + wv[k].x;  sn.y = first_vec ? wv[k].y : b1 * sv[k].y + wv[k].y;
        xn.x = xv[k].x + a0 * pn.x;  xn.y = xv[k].y + a1 * pn.y;
        rn.x = rv[k].x - a0 * sn.x;  rn.y = rv[k].y - a1 * sn.y;
        if (!on0) { pn.x = pv[k].x; sn.x = sv[k].x; xn.x = xv[k].x; rn.x = rv[k].x; }
        if (!on1) { pn.y = pv[k].y; sn.y = sv[k].y; xn.y = xv[k].y; rn.y = rv[k].y; }
        if (on0 || on1) { p2o[g] = pn; s2o[g] = sn; x2o[g] = xn; r2o[g] = rn; }
    }
}
// [part][n][3] <-> [n][6]
__global__ void k_pack6(int n, const double* __restrict__ x, double* __restrict__ x6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n * 3) return;
    const int c = i % 3, part = i / (n * 3), px = (i - part * n * 3) / 3;
    x6[(size_t)px * 6 + part * 3 + c] = x[i];
}
__global__ void k_unpack6(int n, const double* __restrict__ x6, double* __restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n * 3) return;
    const int c = i % 3, part = i / (n * 3), px = (i - part * n * 3) / 3;
    x[i] = x6[(size_t)px * 6 + part * 3 + c];
}

// x: [2][n][3] in (the local-statistics guess) / out (iterate at the cap). gx, gy: gradient weights of the level (k_gradient_weights, lambda = local weight).
int nctk_s1_solve(nct_ctx* ctx, hipStream_t s, const nct_s1_graph& g, const int* knn_id, const double* weight, float dWeight, const uint8_t* s_lab_level,
                  const uint8_t* g_lab_level, const double* gx, const double* gy, int layer, int h, int w, double* x, int* cg_iters_host /*nullable: [3], synchronises*/) {
    const int n = h * w, nbl = cdiv(n, 256);
    DevBuf<double> daa(ctx, (size_t)3 * n), dab(ctx, (size_t)3 * n), dbb(ctx, (size_t)3 * n), rhs(ctx, (size_t)6 * n);
    DevBuf<double> x6(ctx, (size_t)6 * n), r6(ctx, (size_t)6 * n), w6(ctx, (size_t)6 * n), p6(ctx, (size_t)6 * n), s6(ctx, (size_t)6 * n), partial(ctx, (size_t)nbl * 6);
    DevBuf<S1State> st(ctx, 2);
    if (!daa.ok() || !dab.ok() || !dbb.ok() || !rhs.ok() || !x6.ok() || !r6.ok() || !w6.ok() || !p6.ok() || !s6.ok() || !partial.ok() || !st.ok()) return NCT_ERR_HIP;
    hipLaunchKernelGGL(k_s1_setup, dim3(nbl), dim3(256), 0, s, n, weight, dWeight, s_lab_level, g_lab_level, (double*)daa, (double*)dab, (double*)dbb, (double*)rhs); LCHK();
    // XCD-aware block order for the operator pass of the bandwidth-bound levels: an XCD's workgroups take a contiguous eighth of the image, so the rows above and below a
    // block's pixels and the block's own records are in ITS L2 (round 4 measured no gain on the two-pass recurrence's operator; on this one: 9.8 -> 9.3-9.5 ms for the
    // finest level of the bench pair, same bits — the partial sums keep their logical block slot). A cluster-major pixel order on top of it (scripts/s1_cluster_probe.py)
    // LOSES: only 41 % of the bench pair's kNN edges stay inside the pixel's own k-means cluster.
    static const int s1_xcd = [] { const char* e = getenv("NCT_S1_XCD"); return e ? atoi(e) : 1; }();
    static const int s1_one_max = [] { const char* e = getenv("NCT_S1_ONE_XCD_MAX"); return e ? atoi(e) : 32; }();    // largest grid (workgroups) that runs on one XCD; 0: off
    const int one_xcd = (nbl <= s1_one_max && nbl <= S1_FUSE_NB) ? 1 + (ctx->home_xcd & 7) : 0;
    const int ogrid = one_xcd ? nbl * 8 : nbl;
    S1Sys S{n, h, w, daa, dab, dbb, gx, gy, knn_id, g, (s1_xcd && n >= 100000) ? 1 : 0, one_xcd};
    const double tol2 = 1e-6 * 1e-6;
    const int maxit = layer == 4 ? 50 : 100;                       // ColorTransfer.cpp:916-921
    const bool coop = n >= 100000;                                  // shared in-edge gathers pay off on the bandwidth-bound levels only
    const bool fused = nbl <= S1_FUSE_NB;
    // hub pass: only where the host knows (or cannot exclude) that the level has in-edge lists longer than one block
    // (unknown — the host has not seen the count yet: a grid that covers the most blocks the level can have, (8 n - 1) / 64, four per workgroup, at most 256 workgroups)
    const int hub_blind = cdiv(cdiv(8 * n, S1_SEG), 4) < 256 ? cdiv(cdiv(8 * n, S1_SEG), 4) : 256;
    const int hub_grid = g.nseg_hint == 0 ? 0 : (g.nseg_hint > 0 ? (cdiv(g.nseg_hint, 4) < 4096 ? cdiv(g.nseg_hint, 4) : 4096) : hub_blind);
    // the second level only where some pixel has more than 64 hub blocks (in-degree above 4160), by the same rule
    const int hub2_grid = (g.nseg_hint == 0 || g.nsup_hint == 0) ? 0 : (g.nsup_hint > 0 ? (cdiv(g.nsup_hint, 4) < 1024 ? cdiv(g.nsup_hint, 4) : 1024) : (cdiv(hub_blind, S1_SEG) < 64 ? cdiv(hub_blind, S1_SEG) : 64));
    auto hub = [&](const double* v) -> int {
        if (hub_grid) { hipLaunchKernelGGL(k_s1_hub, dim3(hub_grid), dim3(256), 0, s, g, v); LCHK(); }
        if (hub2_grid) { hipLaunchKernelGGL(k_s1_hub2, dim3(hub2_grid), dim3(256), 0, s, g); LCHK(); }
        return 0;
    };
    auto apply = [&](bool kt) -> int {
        if (kt) { int rk = ctx->kt_begin(s, NCT_KT_S1_APPLY); if (rk) return rk; }
        if (coop) hipLaunchKernelGGL(k_s1_apply<true>, dim3(nbl), dim3(256), 0, s, S, (const double*)r6, (double*)w6, (double*)partial);
        else      hipLaunchKernelGGL(k_s1_apply<false>, dim3(ogrid), dim3(256), 0, s, S, (const double*)r6, (double*)w6, (double*)partial);
        LCHK();
        if (kt) { int rk = ctx->kt_end(s); if (rk) return rk; }
        return 0;
    };
    hipLaunchKernelGGL(k_pack6, dim3(cdiv(6 * n, 256)), dim3(256), 0, s, n, (const double*)x, (double*)x6); LCHK();
    { int rc = hub(x6); if (rc) return rc; }
    if (coop) hipLaunchKernelGGL(k_s1_residual<true>, dim3(nbl), dim3(256), 0, s, S, (const double*)x6, (const double*)rhs, (double*)r6);
    else      hipLaunchKernelGGL(k_s1_residual<false>, dim3(nbl), dim3(256), 0, s, S, (const double*)x6, (const double*)rhs, (double*)r6);
    LCHK();
    { int rc = hub(r6); if (rc) return rc; rc = apply(false); if (rc) return rc; }
    // ST_j = state after operator pass j lives in slot j & 1; vector pass k uses ST_{k-1}
    S1State* slot[2] = {(S1State*)st, (S1State*)st + 1};
    for (int k = 1; k <= maxit; ++k) {
        const bool kt = ctx->kt_on && layer == 4 && k >= 3 && k < 11;          // NCT_FLAG_TIME_KERNELS: eight iterations of the finest level, one event pair per launch
        if (fused) {
            hipLaunchKernelGGL(k_s1_update<true>, dim3(ogrid), dim3(256), 0, s, n, nbl, (const double*)partial, (const S1State*)slot[k & 1], slot[(k - 1) & 1], k == 1 ? 1 : 0, tol2,
                               k == 1 ? 1 : 0, (double*)r6, (const double*)w6, (double*)p6, (double*)s6, (double*)x6, one_xcd); LCHK();
        } else {
            if (kt) { int rk = ctx->kt_begin(s, NCT_KT_S1_SCALARS); if (rk) return rk; }
            hipLaunchKernelGGL(k_s1_scal, dim3(1), dim3(256), 0, s, (const double*)partial, nbl, (const S1State*)slot[k & 1], slot[(k - 1) & 1], k == 1 ? 1 : 0, tol2); LCHK();
            if (kt) { int rk = ctx->kt_end(s); if (rk) return rk; rk = ctx->kt_begin(s, NCT_KT_S1_UPDATE); if (rk) return rk; }
            hipLaunchKernelGGL(k_s1_update<false>, dim3(nbl), dim3(256), 0, s, n, nbl, (const double*)partial, (const S1State*)slot[(k - 1) & 1], slot[(k - 1) & 1], 0, tol2,
                               k == 1 ? 1 : 0, (double*)r6, (const double*)w6, (double*)p6, (double*)s6, (double*)x6, 0); LCHK();
            if (kt) { int rk = ctx->kt_end(s); if (rk) return rk; }
        }
        if (k < maxit) { int rc = hub(r6); if (rc) return rc; rc = apply(kt); if (rc) return rc; }
    }
    hipLaunchKernelGGL(k_unpack6, dim3(cdiv(6 * n, 256)), dim3(256), 0, s, n, (const double*)x6, x); LCHK();
    if (cg_iters_host) {
        S1State hst;
        NCT_HIP(hipMemcpyAsync(&hst, slot[(maxit - 1) & 1], sizeof hst, hipMemcpyDeviceToHost, s));
        NCT_HIP(hipStreamSynchronize(s));
        for (int c = 0; c < 3; ++c) cg_iters_host[c] = hst.iters[c];
    }
    return 0;
}
