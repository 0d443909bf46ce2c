"""Dev tool: run bench.py against the diagnostic library (environment switches active): python tools/bench_dbg.py <bench args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from facerecognition_infrenceengine_amd import _lib
_lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), "libfrhip_debug.so"))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
