import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from oracle import oracle as O
from video import ops, _hip
from video.engine import FrameEngine
O.build()
def run(shape, sigma, rate, k, seed, hook):
    rng = np.random.default_rng(seed)
    n = shape[0]; c = shape[3] if len(shape) == 4 else 1
    h, w = shape[1], shape[2]
    clip = (rng.random(shape, dtype=np.float32) * 2 - 0.5).astype(np.float32)
    _hip.check(_hip.lib().va_test_hook_gaussian_f32(hook))
    eng = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema", bg_rate=rate, sigma=sigma)
    parts = [eng.run(p, want=("filtered",))["filtered"] for p in (clip[:k], clip[k:]) if len(p)]
    got = np.concatenate(parts)
    state, seen = eng.get_background(); eng.close()
    _hip.check(_hip.lib().va_test_hook_gaussian_f32(0))
    diff, bg = O.bg_ema_f32(clip.reshape(n, -1), rate=np.float32(rate))
    ref = O.gaussian_f32(diff.reshape(shape), sigma)
    bad = got.view(np.uint32) != ref.view(np.uint32)
    sbad = state.ravel().view(np.uint32) != bg.ravel().view(np.uint32)
    return int(bad.sum()), int(sbad.sum()), (np.argwhere(bad)[:5].tolist() if bad.any() else [])
for hook in (0, 1, 2, 3):
    for seed in range(6):
        for k in (0, 1):
            r = run((2, 379, 1280), 1.0, 0.3, k, seed, hook)
            if r[0] or r[1]:
                print("hook", hook, "seed", seed, "split", k, r)
print("done")
