"""PatchMatch fixture only (finest level of BASELINE config 2): used under rocprofv3 --pmc to read FETCH_SIZE / WRITE_SIZE."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 700
c = nct.Context(0)
c.pm_bench_setup(synth.features(11, 64, S, S), synth.features(12, 64, S, S))
for d in range(2):
    ms, _, _, _ = c.pm_bench_run(iters=10, rs_max=32, seed=17 + d)
    print("pass", d, "kernel ms", ms)
