"""Run-to-run bit equality of the uint8 chain (labels, counts, mask, blur) at full size.  On an MI355X:
    python tools/debug/determinism_u8.py [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from video.engine import FrameEngine  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
bad = 0
for (n, h, w, sigma) in ((256, 1080, 1920, 5.0), (64, 2160, 3840, 5.0), (256, 1080, 1920, 8.0)):
    g = torch.Generator(device=dev); g.manual_seed(n + h)
    base = torch.randint(90, 110, (1, h, w), generator=g, device=dev, dtype=torch.int16)
    x = (base + torch.randint(-4, 5, (n, h, w), generator=g, device=dev, dtype=torch.int16)).clamp(0, 255).to(torch.uint8)
    yy = torch.arange(h, device=dev).view(1, h, 1); xx = torch.arange(w, device=dev).view(1, 1, w)
    for k in range(40):
        cx, cy, r = (97 * k * 7) % w, (61 * k * 5) % h, 8 + k
        x[:, :, :] = torch.where(((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r, (x.to(torch.int16) + 60).clamp(0, 255).to(torch.uint8), x)
    eng = FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=sigma, thresh=20,
                      morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4, device=0)
    lab = [torch.empty((n, h, w), dtype=torch.int32, device=dev) for _ in range(2)]
    cnt = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(2)]
    flt = [torch.empty((n, h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
    state0 = np.full((h, w), 100.0)
    eng.set_background(state0, 50)
    eng.run_device(x.data_ptr(), n, flt[0].data_ptr(), None, lab[0].data_ptr(), cnt[0].data_ptr(), None, st)
    torch.cuda.synchronize()
    d = 0
    for r in range(runs):
        eng.set_background(state0, 50)
        eng.run_device(x.data_ptr(), n, flt[1].data_ptr(), None, lab[1].data_ptr(), cnt[1].data_ptr(), None, st)
        torch.cuda.synchronize()
        d += int((lab[1] != lab[0]).sum().item()) + int((cnt[1] != cnt[0]).sum().item()) + int((flt[1] != flt[0]).sum().item())
    print("%s: %d frames of %dx%d, sigma %g, %d objects in frame 0: %d differing values over %d repeated runs"
          % (eng.description, n, w, h, sigma, int(cnt[0][0].item()), d, runs), flush=True)
    eng.close()
    bad += d
    del lab, flt, x
sys.exit(1 if bad else 0)
