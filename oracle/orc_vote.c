/* oracle/orc_vote.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md). "parity unpinned".
 *
 * CPU restatement of the bidirectional-similarity (BDS) votes:
 *   B1  reconstruct_bds   GeneralizedPatchMatch.cu:122-235   (image domain, u8, host code in the reference)
 *   B2  avg_vote_bds_a    GeneralizedPatchMatch.cu:1074-1126 (coherence, gather)
 *       avg_vote_bds_b    GeneralizedPatchMatch.cu:1128-1178 (completeness, float atomicAdd scatter)
 *       avg_vote_bds      GeneralizedPatchMatch.cu:1180-1202 (divide by weight)
 *
 * Divergences (SPEC.md):
 *  - vote_weight is zeroed first (the reference does `pw[aid] += wa` on an uninitialised cudaMalloc, main.cu:299).
 *  - the completeness scatter's atomic order is nondeterministic in the reference; any fixed order is one of its legal realisations. Canonical order v2 (round 5; rounds
 *    1-4 added in ascending source pixel across all taps): AFTER all coherence contributions (kernel a completes before kernel b starts), TAP-MAJOR: for the nine
 *    taps in the kernel's own loop order (dx outer, dy inner), the sources whose match is the tapped neighbour s = target - tap, in ascending source pixel. A list of
 *    more than ORC_VOTE_SEG (64) sources — natural photographs collapse 10^4 pixels of a flat region onto one match (demo/example/in/in4.png: 47 535 sources on one
 *    target), a 47 535-step chain of float adds — is added in blocks: its first 64 sources one by one into the target's sums, every further block of 64 (ascending,
 *    relative to the list's start) summed from zero in the same way and then added as ONE addend, in block order. pw follows the same order.
 * Mixed precision is kept literally: `pw += wa` and `pout += pin*wa` are float += double (evaluated in double,
 * rounded to float); atomicAdd(float*, double-expression) rounds the addend to float first, then adds in float.
 */
#include "orc_common.h"
#define ORC_VOTE_SEG 64

/* B2 — features are CHW fp32; pin = UN-normalised R features (C,bh,bw); pout = voted features (C,ah,aw). */
void orc_bds_vote_features(const uint32_t* ann, const uint32_t* bnn, const float* pin, float* pout, float* pw_out /*nullable*/,
                           int C, int ah, int aw, int bh, int bw, int patch, float wCohen, float wComplete) {
    int slice_a = ah * aw, slice_b = bh * bw, r = patch / 2;
    float* pw = (float*)calloc(slice_a, sizeof(float));
    double wa = wCohen / (double)(aw * ah);
    double wb = wComplete / (double)(bw * bh);
    /* kernel a */
#pragma omp parallel for schedule(static)
    for (int ay = 0; ay < ah; ++ay)
        for (int ax = 0; ax < aw; ++ax) {
            int aid = ay * aw + ax;
            for (int c = 0; c < C; ++c) pout[(size_t)c * slice_a + aid] = 0;
            for (int dx = -r; dx <= r; ++dx)
                for (int dy = -r; dy <= r; ++dy) {
                    if ((ax + dx) < aw && (ax + dx) >= 0 && (ay + dy) < ah && (ay + dy) >= 0) {
                        uint32_t vp = ann[(ay + dy) * aw + ax + dx];
                        int xp = orc_int_to_x(vp) - dx, yp = orc_int_to_y(vp) - dy;
                        if (xp < bw && xp >= 0 && yp < bh && yp >= 0) {
                            pw[aid] = (float)((double)pw[aid] + wa);
                            for (int c = 0; c < C; ++c) {
                                double t = (double)pin[(size_t)c * slice_b + yp * bw + xp] * wa;
                                pout[(size_t)c * slice_a + aid] = (float)((double)pout[(size_t)c * slice_a + aid] + t);
                            }
                        }
                    }
                }
        }
    /* kernel b — canonical order v2: gather form over the inverse of the R->S field (lists of sources per matched S pixel, ascending source pixel) */
    float wbf = (float)wb;
    int* start = (int*)calloc((size_t)slice_a + 1, sizeof(int));
    int* list = (int*)malloc(sizeof(int) * (size_t)slice_b);
    for (int q = 0; q < slice_b; ++q) { uint32_t vp = bnn[q]; start[orc_int_to_y(vp) * aw + orc_int_to_x(vp) + 1]++; }
    for (int s = 0; s < slice_a; ++s) start[s + 1] += start[s];
    { int* cur = (int*)malloc(sizeof(int) * (size_t)slice_a); memcpy(cur, start, sizeof(int) * (size_t)slice_a);
      for (int q = 0; q < slice_b; ++q) { uint32_t vp = bnn[q]; list[cur[orc_int_to_y(vp) * aw + orc_int_to_x(vp)]++] = q; }
      free(cur); }
#pragma omp parallel for schedule(dynamic, 16)
    for (int aid = 0; aid < slice_a; ++aid) {
        const int ay = aid / aw, ax = aid - ay * aw;
        float* blk = (float*)malloc(sizeof(float) * (size_t)C);
        for (int dx = -r; dx <= r; ++dx)
            for (int dy = -r; dy <= r; ++dy) {
                const int sx = ax - dx, sy = ay - dy;                   /* the matched pixel whose sources reach this target through tap (dx, dy) */
                if (sx < 0 || sx >= aw || sy < 0 || sy >= ah) continue;
                const int e0 = start[sy * aw + sx], e1 = start[sy * aw + sx + 1];
                for (int b0 = e0; b0 < e1; b0 += ORC_VOTE_SEG) {
                    const int first = b0 == e0, b1 = b0 + ORC_VOTE_SEG < e1 ? b0 + ORC_VOTE_SEG : e1;
                    float bw_acc = 0.f;
                    if (!first) for (int c = 0; c < C; ++c) blk[c] = 0.f;
                    for (int e = b0; e < b1; ++e) {
                        const int q = list[e], by = q / bw, bx = q - by * bw;
                        const int xb = bx + dx, yb = by + dy;
                        if (xb < bw && xb >= 0 && yb < bh && yb >= 0) {
                            const int bid = yb * bw + xb;
                            if (first) pw[aid] = pw[aid] + wbf; else bw_acc = bw_acc + wbf;
                            for (int c = 0; c < C; ++c) {
                                float t = (float)(wb * (double)pin[(size_t)c * slice_b + bid]);
                                if (first) pout[(size_t)c * slice_a + aid] = pout[(size_t)c * slice_a + aid] + t; else blk[c] = blk[c] + t;
                            }
                        }
                    }
                    if (!first) {
                        pw[aid] = pw[aid] + bw_acc;
                        for (int c = 0; c < C; ++c) pout[(size_t)c * slice_a + aid] = pout[(size_t)c * slice_a + aid] + blk[c];
                    }
                }
            }
        free(blk);
    }
    free(start); free(list);
    /* avg_vote_bds */
#pragma omp parallel for schedule(static)
    for (int aid = 0; aid < slice_a; ++aid)
        if (pw[aid] > 0)
            for (int c = 0; c < C; ++c) pout[(size_t)c * slice_a + aid] /= pw[aid];
    if (pw_out) memcpy(pw_out, pw, sizeof(float) * slice_a);
    free(pw);
}

/* B1 — reconstruct_bds: a,b are u8 BGR HWC images (level-size S and R); result G has a's size.
 * Integer accumulators => order-independent, exact. Final blend in double, truncating conversion to u8. */
void orc_bds_vote_image(const uint8_t* a, int ah, int aw, const uint8_t* b, int bh, int bw,
                        const uint32_t* ann, const uint32_t* bnn, int patch, double wCohen, double wComplete, uint8_t* out) {
    int n = ah * aw;
    int* aRes = (int*)calloc((size_t)n * 3, sizeof(int));
    int* bRes = (int*)calloc((size_t)n * 3, sizeof(int));
    int* aWgt = (int*)calloc(n, sizeof(int));
    int* bWgt = (int*)calloc(n, sizeof(int));
    int leftSize = -patch / 2, rightSize = patch + leftSize - 1;
    double wa = wCohen / (double)(aw * ah);
    double wb = wComplete / (double)(bw * bh);
    for (int ay = 0; ay < ah; ++ay)
        for (int ax = 0; ax < aw; ++ax) {
            int cnt = 0, col[3] = {0, 0, 0};
            for (int dx = leftSize; dx <= rightSize; ++dx)
                for (int dy = leftSize; dy <= rightSize; ++dy)
                    if ((ax + dx) < aw && (ax + dx) >= 0 && (ay + dy) < ah && (ay + dy) >= 0) {
                        uint32_t vp = ann[(ay + dy) * aw + ax + dx];
                        int xp = orc_int_to_x(vp), yp = orc_int_to_y(vp);
                        int nx = xp - dx, ny = yp - dy;
                        if (nx < bw && nx >= 0 && ny < bh && ny >= 0) {
                            const uint8_t* bv = b + ((size_t)ny * bw + nx) * 3;
                            col[0] += bv[0]; col[1] += bv[1]; col[2] += bv[2];
                            cnt++;
                        }
                    }
            int id = ay * aw + ax;
            aRes[id * 3 + 0] += col[0]; aRes[id * 3 + 1] += col[1]; aRes[id * 3 + 2] += col[2];
            aWgt[id] += cnt;
        }
    int r = patch / 2;
    for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
            uint32_t vp = bnn[by * bw + bx];
            int xp = orc_int_to_x(vp), yp = orc_int_to_y(vp);
            for (int dx = -r; dx <= r; ++dx)
                for (int dy = -r; dy <= r; ++dy)
                    if ((bx + dx) < bw && (bx + dx) >= 0 && (by + dy) < bh && (by + dy) >= 0)
                        if ((xp + dx) < aw && (xp + dx) >= 0 && (yp + dy) < ah && (yp + dy) >= 0) {
                            int id = (yp + dy) * aw + xp + dx;
                            const uint8_t* bv = b + ((size_t)(by + dy) * bw + bx + dx) * 3;
                            bRes[id * 3 + 0] += bv[0]; bRes[id * 3 + 1] += bv[1]; bRes[id * 3 + 2] += bv[2];
                            bWgt[id] += 1;
                        }
        }
    for (int id = 0; id < n; ++id) {
        double awt = aWgt[id] * wa, bwt = bWgt[id] * wb;
        for (int c = 0; c < 3; ++c) {
            double v = (double)(aRes[id * 3 + c] * wa + bRes[id * 3 + c] * wb) / (double)(awt + bwt);
            out[id * 3 + c] = (uint8_t)v;     /* C++ implicit double -> uchar: truncation */
        }
    }
    (void)a;
    free(aRes); free(bRes); free(aWgt); free(bWgt);
}
