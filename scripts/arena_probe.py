"""Device memory a context's arena holds after one pair (NCT_CTR_ARENA_BYTES): the high-water mark per image size.   usage: python scripts/arena_probe.py [sizes ...]"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
ws, bs = synthetic_vgg19(19)
for S in [int(a) for a in sys.argv[1:]] or [256, 700, 1000]:
    with nct.Context(0) as c:
        c.vgg19_load_raw(ws, bs)
        c.pair_upload(synth.image(1000, S, S), synth.image(1001, S, S))
        c.pair_run(nct.Params.default())
        b1 = c.counter(nct.CTR_ARENA_BYTES)
        c.pair_run(nct.Params.default())
        print(f"{S}x{S}: arena {b1 / 1e9:.3f} GB after one pair, {c.counter(nct.CTR_ARENA_BYTES) / 1e9:.3f} GB after two = {b1 / (S * S):.0f} B per pixel", flush=True)
