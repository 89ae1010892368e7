"""VGG19 forward only (700x700 -> conv5_1, synthetic weights): used under rocprofv3 --pmc for the MFMA counters of k_conv3x3_mfma2b."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
c = nct.Context(0)
ws, bs = synthetic_vgg19(19)
c.vgg19_load_raw(ws, bs)
img = synth.image(1000, 700, 700)
for _ in range(3):
    c.vgg19_features(img, 5)
