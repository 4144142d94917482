"""Run-to-run bit equality of the float32 blur (plain and with the EMA background) at full size: a rare store-data
corruption shows up as a difference between repeated runs of the same input.  On an MI355X:
    python tools/debug/determinism_f32.py [runs [lib_<variant>.so]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from video import _hip  # noqa: E402
from video.engine import FrameEngine  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if len(sys.argv) > 2:                       # another build of the library: video-analysis_amd/csrc/build/dbg/<name>
    _hip.LIB_PATH = os.path.join(ROOT, "video-analysis_amd/csrc/build/dbg", sys.argv[2])
L = _hip.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
bad = 0
for (n, h, w, c, sigma) in ((64, 1080, 1920, 3, 9.0), (64, 1080, 1920, 1, 5.0), (32, 2160, 3840, 1, 3.0), (48, 583, 1920, 3, 5.0)):
    g = torch.Generator(device=dev); g.manual_seed(n + c)
    shape = (n, h, w, c) if c == 3 else (n, h, w)
    x = torch.rand(shape, generator=g, device=dev, dtype=torch.float32)
    ref = torch.empty_like(x); out = torch.empty_like(x)
    _hip.check(L.va_gaussian_f32(x.data_ptr(), ref.data_ptr(), n, h, w, c, sigma, st)); torch.cuda.synchronize()
    d = 0
    for r in range(runs):
        _hip.check(L.va_gaussian_f32(x.data_ptr(), out.data_ptr(), n, h, w, c, sigma, st)); torch.cuda.synchronize()
        d += int((out.view(torch.int32) != ref.view(torch.int32)).sum().item())
    print("va_gaussian_f32 %s sigma=%g: %d differing samples over %d repeated runs" % (shape, sigma, d, runs), flush=True)
    bad += d
    eng = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema", bg_rate=0.02, sigma=sigma, device=0)
    s0 = np.zeros((h, w, c) if c == 3 else (h, w), np.float32)
    eng.set_background(s0, 5)
    eng.run_device(x.data_ptr(), n, ref.data_ptr(), None, None, None, None, st); torch.cuda.synchronize()
    d = 0
    for r in range(runs):
        eng.set_background(s0, 5)
        eng.run_device(x.data_ptr(), n, out.data_ptr(), None, None, None, None, st); torch.cuda.synchronize()
        d += int((out.view(torch.int32) != ref.view(torch.int32)).sum().item())
    eng.close()
    print("EMA + blur      %s sigma=%g: %d differing samples over %d repeated runs" % (shape, sigma, d, runs), flush=True)
    bad += d
sys.exit(1 if bad else 0)
