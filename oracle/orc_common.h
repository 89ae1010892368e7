/* oracle/orc_common.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 * Shared helpers for the CPU restatement of the reference hot path. */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#ifdef __cplusplus
extern "C" {
#endif

/* NNF element packing — GeneralizedPatchMatch.cu:24-34 (XY_TO_INT / INT_TO_X / INT_TO_Y). */
static inline uint32_t orc_xy_to_int(int x, int y) { return ((uint32_t)y << 12) | (uint32_t)x; }
static inline int orc_int_to_x(uint32_t v) { return (int)(v & 0xFFFu); }
static inline int orc_int_to_y(uint32_t v) { return (int)((v >> 12) & 0xFFFu); }
/* GeneralizedPatchMatch.cu:9-22 — note argument order (x, x_max, x_min). */
static inline int orc_clamp(int x, int x_max, int x_min) { return x > x_max ? x_max : (x < x_min ? x_min : x); }

/* Counter-based RNG replacing cuRAND XORWOW (GeneralizedPatchMatch.cu:54-66; curand_init(seed = global x index)
 * is not reproducible here and the reference schedule is racy — documented divergence, DESIGN.md §4.4, SPEC.md).
 * Returns u in (0,1] like curand_uniform. */
static inline uint32_t orc_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
static inline float orc_rand_u01(uint32_t seed, int ax, int ay, int iter, int step, int axis) {
    uint32_t ctr = (uint32_t)(1 + iter * 64 + step * 2 + axis);
    uint32_t h = orc_mix32(seed ^ orc_mix32((uint32_t)(ay * 4096 + ax) + 0x9E3779B9u * ctr));
    return (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
}

#ifdef __cplusplus
}
#endif
#endif
