# kernel trace (no counters) of three VGG19 forwards at 700x700: real per-layer durations.  usage: bash scripts/vgg_trace.sh <tag>
out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
for lib in "" $(ls neural-color-transfer_amd/lib/variants/*.so 2>/dev/null); do
  name=default; [ -n "$lib" ] && name=$(basename $lib .so) && export NCT_LIB=$PWD/$lib
  timeout 300 rocprofv3 --kernel-trace -d $out/vggtrace_$name -o t --output-format csv -- python scripts/vgg_only.py > $out/vggtrace_$name.log 2>&1
  python - $out/vggtrace_$name/t_kernel_trace.csv $name <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    d[(n.split('(')[0][:40], int(r['Grid_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in sorted(d.items()):
    v = v[len(v) // 3:]            # drop the first forward (cold)
    print(sys.argv[2], k, len(v), 'avg us %.1f' % (sum(v) / len(v)), 'sum %.1f' % sum(v)); tot += sum(v)
print(sys.argv[2], 'us per forward', tot / 2)
PY
done
