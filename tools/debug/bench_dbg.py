import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from video import _hip
_hip.LIB_PATH = os.path.join(ROOT, "video-analysis_amd/csrc/build/dbg", sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
