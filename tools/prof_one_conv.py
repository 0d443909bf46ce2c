"""Dev tool: 20 launches of the stage-3 body conv (14x14, 256->256, 256 faces) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import bench_conv
bench_conv.run(256, 14, 14, 256, 256, iters=20, tag="pmc")
