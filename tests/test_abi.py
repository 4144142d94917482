"""CPU: the C-ABI library loads without a GPU, exports every symbol include/*.h declares, the
ctypes prototypes cover the same set, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "videoanalysis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(va_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from video import _hip
    lib = _hip.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(_hip.SIGNATURES) == declared


def test_config_struct_matches_header_layout():
    from video import _hip
    # 6 int32, int32 bg_mode, float, double (8-aligned), 3 int32, 3x4 int32, 2 int32
    assert C.sizeof(_hip.va_config) == 112
    assert _hip.va_config.sigma.offset == 32 and _hip.va_config.thresh.offset == 40


def test_taps_need_no_gpu_and_match_oracle(golden, oracle):
    from video import _hip
    for s in (0.5, 1.0, 2.0, 3.0, 5.0):
        assert np.array_equal(_hip.gauss_taps_q8(s), golden["taps_q8_%g" % s])
        assert np.array_equal(_hip.gauss_taps_q8(s), oracle.gauss_taps_q8(s))
    for s in (2.0, 9.0):
        assert np.array_equal(_hip.gauss_taps_f32(s), oracle.gauss_taps_f32(s))
    with pytest.raises(_hip.HipError):
        _hip.gauss_taps_q8(-1.0)
    with pytest.raises(_hip.HipError):
        _hip.gauss_taps_q8(100.0)          # 601 taps > 255 supported


def test_product_path_fails_loudly_without_gpu():
    from video import _hip
    if _hip.gpu_available():
        pytest.skip("a GPU is present")
    from video import ops
    from video.filters import FilterBlur
    from video.io.memory import VideoMemory
    with pytest.raises(_hip.HipUnavailableError):
        ops.gaussian_blur(np.zeros((8, 8), np.uint8), 2.0)
    v = FilterBlur(VideoMemory(np.zeros((2, 8, 8), np.uint8)), 2)
    with pytest.raises(_hip.HipUnavailableError):
        next(iter(v))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "video-analysis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "va_oracle" not in text and "import oracle" not in text \
                    and "from oracle" not in text, os.path.join(dirpath, f)
