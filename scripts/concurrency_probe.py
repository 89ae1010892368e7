"""How much does the per-step join of bench.py cost? K worker threads (own contexts) run `n` resident 700x700 pairs each, (a) joined after every pair (bench.py's
steps), (b) back to back without joins, (c) back to back with the workers started a quarter of a pair apart. usage: concurrency_probe.py [K] [n]"""
import sys, time, threading
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ws, bs = synthetic_vgg19(19)
ctxs = [nct.Context(0) for _ in range(K)]
for k, c in enumerate(ctxs):
    c.vgg19_load_raw(ws, bs)
    c.pair_upload(synth.image(1000 + 2 * k, 700, 700), synth.image(1001 + 2 * k, 700, 700))
    c.pair_run()
prm = nct.Params.default()

def run(fn):
    ths = [threading.Thread(target=fn, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    return time.perf_counter() - t0

t = 0.0
for i in range(n):
    t += run(lambda k: ctxs[k].pair_run(prm))
print(f"(a) joined per pair      : {K * n / t:.2f} pairs/s")
t = run(lambda k: [ctxs[k].pair_run(prm) for _ in range(n)])
print(f"(b) back to back         : {K * n / t:.2f} pairs/s")
def stag(k):
    time.sleep(0.030 * k)
    for _ in range(n): ctxs[k].pair_run(prm)
t = run(stag)
print(f"(c) staggered by 30 ms   : {K * n / t:.2f} pairs/s (includes the stagger itself)")
