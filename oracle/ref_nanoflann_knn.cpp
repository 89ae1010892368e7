// oracle/ref_nanoflann_knn.cpp — TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, only where /root/reference is mounted).
//
// Drives the reference's OWN vendored KD-tree library (ColorTransfer/Flann/nanoflann.hpp, included from where it lies — nothing
// of the reference is copied into this repository) exactly the way ColorTransfer::findSubKNNs does (ColorTransfer.cpp:136-190):
// 3-D Lab points of one cluster, L2_Simple_Adaptor over a data source whose kdtree_distance returns the EUCLIDEAN distance
// (sqrt, clamped at 0 — ColorTransfer.cpp:20-27), KDTreeSingleIndexAdaptorParams(k) leaves, knnSearch for k+1 results per point.
// The output pins oracle/orc_color.c's brute-force kNN (and through it the GPU colour-grid search): tests/golden/gen_knn_nanoflann.py
// writes the fixture, tests/test_oracle_color.py compares.
//
// usage: ref_nanoflann_knn <in.bin> <out.bin>
//   in : int32 n, int32 k, n*3 float64 points            out: n*(k+1) int32 indices (-1 = none), n*(k+1) float64 distances
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
using std::max;
#include "nanoflann.hpp"

struct Cloud {
    std::vector<double> xyz;                                             // n x 3
    inline size_t kdtree_get_point_count() const { return xyz.size() / 3; }
    inline double kdtree_distance(const double* p, const int idx, int /*size*/) const {
        const double d0 = p[0] - xyz[3 * (size_t)idx], d1 = p[1] - xyz[3 * (size_t)idx + 1], d2 = p[2] - xyz[3 * (size_t)idx + 2];
        return std::max(std::sqrt(d0 * d0 + d1 * d1 + d2 * d2), 0.0);
    }
    inline double kdtree_get_pt(const int idx, int dim) const { return xyz[3 * (size_t)idx + dim]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("in"); return 1; }
    int32_t n = 0, k = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&k, 4, 1, f) != 1 || n <= 0 || k <= 0) { fprintf(stderr, "bad header\n"); return 1; }
    Cloud cloud; cloud.xyz.resize((size_t)n * 3);
    if (fread(cloud.xyz.data(), 8, (size_t)n * 3, f) != (size_t)n * 3) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, Cloud>, Cloud, 3, int> tree_t;
    tree_t index(3, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(k));
    index.buildIndex();
    const size_t m = (size_t)k + 1;
    std::vector<int32_t> ids((size_t)n * m, -1);
    std::vector<double> ds((size_t)n * m, 1e300);
    for (int s = 0; s < n; ++s) {
        const double q[3] = {cloud.xyz[3 * (size_t)s], cloud.xyz[3 * (size_t)s + 1], cloud.xyz[3 * (size_t)s + 2]};
        std::vector<int> ri(m, -1); std::vector<double> rd(m, 1e300);
        const size_t got = index.knnSearch(q, m, ri.data(), rd.data());
        for (size_t t = 0; t < got; ++t) { ids[(size_t)s * m + t] = ri[t]; ds[(size_t)s * m + t] = rd[t]; }
    }
    f = fopen(argv[2], "wb");
    if (!f) { perror("out"); return 1; }
    fwrite(ids.data(), 4, ids.size(), f); fwrite(ds.data(), 8, ds.size(), f);
    fclose(f);
    return 0;
}
