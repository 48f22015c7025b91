#!/usr/bin/env python3
"""The fused region-attention kernel at B = 1024 (one caption row per image, 36 regions, A = 512, R = 1000) exactly as
bench.attention_large_batch runs it -- rotating over 6 independent input sets (1.4 GB > 5 x the Infinity Cache), then on one
resident set -- as a standalone command, so that `rocprofv3 --kernel-trace --stats` can put its own per-launch duration beside the
in-library HIP-event figure the bench line carries.  Prints the HIP-event result as one JSON line."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    out = bench.attention_large_batch(dev)
    print(json.dumps(out))
