"""bench.py's `projection` leg alone: one GPU does the per-round work of rank 0 of N (FramePipeline(emulate=(0, N))).
usage: python tools/round_emulation.py [N] [--sam none] [--encoder-batch B] ...   (bench.py's flags; env knobs OVO_MAIN_PRIORITY, OVO_SAM_CUS, OVO_VIT_CUS)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if not a.isdigit() or a != str(N)]
args = bench.parse()
args.projection_world = N
from ovo_amd.pipeline import synthetic_frames
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if os.environ.get("OVO_MAIN_PRIORITY"):
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["OVO_MAIN_PRIORITY"])))
frames = synthetic_frames(48, dev)
sam = None if args.sam == "none" else args.sam
print(json.dumps(bench.projection_leg(args, dev, frames, sam)))
