"""bench.py's cfg3 block alone (chunked Xent + whole utterances with Warp-CTC): python devtools/bench_cfg3_block.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import aslp_import
aslp = aslp_import.load()
aslp.ops.use_torch_stream()
b = bench.cfg3_block(aslp, torch.device("cuda:0"))
print("chunked %.4f ms  whole %.3f ms  frac %.3f / %.3f" % (b["chunked_xent"]["ms_per_step"], b["whole_utterance_warpctc"]["ms_per_step"],
      b["chunked_xent"]["frac"], b["whole_utterance_warpctc"]["frac"]))
