# per-kernel totals of ONE 700x700 pair (kernel trace, no counters), filtered by a name pattern.  usage: bash scripts/kernel_times.sh <tag> <grep pattern> [size | natural case]
out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o k -- python scripts/pair_only.py ${3:-700} 2 > $out/kt.log 2>&1
python - $out/kt/k_kernel_stats.csv "$2" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print('%-60s calls %5s total_us %10.1f avg_us %8.1f min %8.1f max %8.1f' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
