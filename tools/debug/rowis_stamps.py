"""Where the waves of one row_is_f32_kernel workgroup spend their cycles (debug build only):

    tools/debug/build_variant.sh va_gauss_f32_fused.hip stamps3 -DROWIS_STAMPS=3
    python tools/debug/rowis_stamps.py lib_stamps3.so [width channels]               # on the GPU box

Per phase (one sub-chunk of one frame) and wave of the middle workgroup: shader-clock cycles waiting in each of the
three barriers (A: row pass and staging done; B: results written, halos fixed; C: copy-out done / compute waves
half-way) and in each part of the phase.  A wave that waits ~0 cycles in a barrier is the one the others wait for.
Waves 0-7 load / stage / copy out, waves 8-15 compute.  (-DROWIS_STAMPS=1: loader waves only, 2: compute waves.)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from video import _hip  # noqa: E402

_hip.LIB_PATH = os.path.join(ROOT, "video-analysis_amd/csrc/build/dbg", sys.argv[1])
import numpy as np  # noqa: E402
import torch  # noqa: E402
from video.engine import FrameEngine  # noqa: E402


def main():
    w, h, c, batch = 1920, 1080, 3, 256
    if len(sys.argv) > 2:
        w, c = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    frames = torch.rand((batch, h, w, c), device=dev, dtype=torch.float32)
    out = torch.empty_like(frames)
    eng = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=batch, background="ema",
                      bg_rate=0.02, sigma=9.0, device=0)
    stream = torch.cuda.current_stream(dev)
    for _ in range(3):
        eng.run_device(frames.data_ptr(), batch, out.data_ptr(), None, None, None, None, stream.cuda_stream)
    torch.cuda.synchronize()
    lib = C.CDLL(_hip.LIB_PATH)
    buf = (C.c_uint * 128)()
    lib.va_debug_rowis_stamps.restype = C.c_int
    assert lib.va_debug_rowis_stamps(buf) == 0
    t = np.array(buf, dtype=np.float64).reshape(16, 8) / (2 * batch)
    print("cycles per phase          wait A   wait B   wait C | loader: stage+loads  fixup  copy-out | "
          "compute: row pass (incl. C)  write O |   total")
    for wv in range(16):
        r = t[wv]
        if r[7] == 0:
            continue
        if wv < 8:
            print("wave %2d load          %8.0f %8.0f %8.0f | %19.0f %6.0f %9.0f |                                       | %7.0f"
                  % (wv, r[0], r[1], r[2], r[3], r[4], r[5], r[7]))
        else:
            print("wave %2d compute       %8.0f %8.0f %8.0f |                                      | %27.0f %8.0f | %7.0f"
                  % (wv, r[0], r[1], r[2], r[3], r[4], r[7]))


if __name__ == "__main__":
    main()
