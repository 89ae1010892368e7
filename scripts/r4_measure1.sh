#!/bin/bash
# round 4, first measurement pass: FETCH_SIZE calibration probe, bench with every live pass, GPU tests of the bench contract
tag=${1:-r4h}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib scripts/probes/fetch_calib.hip || exit 1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $out/calib/pass_$n -o c --output-format csv -- /tmp/fetch_calib > $out/calib_$n.log 2>&1
done
python scripts/probes/fetch_calib_report.py $out/calib > $out/fetch_calibration.md 2>&1; cat $out/fetch_calibration.md
find $out/calib -name "*kernel_trace*" -delete
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > $out/pytest_bench.txt 2>&1; tail -3 $out/pytest_bench.txt
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err; python - $out/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ("value","host_to_host_pairs_per_s","single_pair_ms","single_pair_ms_min")})
print("stages",{k:(round(v,2) if isinstance(v,float) else v) for k,v in d["stages_ms"].items() if not isinstance(v,list)}, d["stages_ms"]["wls_iters"])
print("roofline frac",d["roofline"]["frac"],"traffic",d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("by_step",{k:round(v["frac_of_hbm_peak"],3) for k,v in (d["roofline"]["pmc"] or {}).get("by_step",{}).items()} if d["roofline"].get("pmc") else None)
print("roofline_1000",{k:v for k,v in d.get("roofline_1000",{}).items() if k in ("frac","achieved","avg_launch_us","restream_factor","why")}, {k:round(v["frac_of_hbm_peak"],3) for k,v in d.get("roofline_1000",{}).get("by_step",{}).items()})
for k,v in d["roofline_color"]["kernels"].items(): print(k,{a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("avg_launch_us","avg_us","samples","frac","achieved")})
print({k:d["roofline_color"][k] for k in ("wls_iteration","s1_iteration")})
print("cpu",d.get("cpu_baseline",{}).get("value"), d["vgg_mfma"].get("frac"), d["vgg_mfma"].get("mfma_util"))
PY
