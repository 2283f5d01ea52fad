import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
import numpy as np, oracle
from q3asr import Qwen3ASRModel, make_config, synth
cfg = make_config("tiny"); m = Qwen3ASRModel(cfg)
for kind in ("noise", "speechlike"):
    a = synth.synth_audio(480000, 11, kind)
    got = m.feature_extractor.process(a, 16000)
    r64 = oracle.mel.extract_features(a, "exact", np.float64)
    e = np.abs(got - r64)
    i = np.unravel_index(np.argmax(e), e.shape)
    print(kind, "max", e.max(), "at mel,frame", i, "got", got[i], "ref", r64[i], "frac>5e-5", (e > 5e-5).mean(), "min ref", r64.min())
    # per-mel max error
    pm = e.max(axis=1)
    print("  per-mel max err (first 8, last 8):", pm[:8], pm[-8:])
    print("  per-frame-block max err:", [float(e[:, j:j+300].max()) for j in range(0, 3000, 300)])
