#!/bin/bash
# per-kernel stats of the kNN stage (rocprofv3 --stats of three single pairs) for the library NCT_LIB selects. usage: scripts/knn_all.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/knnall
rocprofv3 --kernel-trace --stats -d gpurun_out/knnall -o t --output-format csv -- python scripts/wls_levels.py 3 > gpurun_out/knnall.log 2>&1
f=$(find gpurun_out/knnall -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("k_knn_") or n.startswith("k_cell_masks"):
        print("%-24s calls %4s  avg %8.1f us  min %8.1f  max %8.1f  total %8.2f ms" % (n.split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 gpurun_out/knnall.log | sed "s/wls_level_ms.*total/total/"
rm -rf gpurun_out/knnall
