#!/bin/bash
# k_knn_grid launch durations per level (rocprofv3 kernel trace of single pairs) for the library NCT_LIB selects. usage: scripts/knn_times.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/knn_$1
rocprofv3 --kernel-trace -d gpurun_out/knn_$1 -o t --output-format csv -- python scripts/wls_levels.py 3 > gpurun_out/knn_$1.log 2>&1
f=$(find gpurun_out/knn_$1 -name "*kernel_trace.csv" | head -1)
test -n "$f" && python - "$f" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    if r["Kernel_Name"].startswith("k_knn_grid"):
        g[int(r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("k_knn_grid us by grid size:", {k: round(sorted(v)[len(v) // 2], 1) for k, v in sorted(g.items())}, "sum %.1f" % sum(sorted(v)[len(v) // 2] for v in g.values()))
PY
tail -1 gpurun_out/knn_$1.log | sed "s/wls_level_ms.*total/total/"
rm -rf gpurun_out/knn_$1
