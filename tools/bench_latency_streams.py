"""Dev tool: single 640x480 frame get() + match latency (bench.c1_latency) by the number of level streams of single-frame calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch, warnings, bench
from facerecognition_infrenceengine_amd import FaceAnalysis
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
for n in (0, 11, 2, 4, 0, 11, 2, 4):
    MTCNNHIP.SINGLE_FRAME_LEVEL_STREAMS = n          # c1_latency clones the engine: the clone's detector reads the class default
    print("single_frame_level_streams", n, bench.c1_latency(app, dev, n=40), flush=True)
