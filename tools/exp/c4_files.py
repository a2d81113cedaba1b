#!/usr/bin/env python3
"""the configs[3] single-GPU leg of bench.py on its own (compare + compare_files end to end); usage: tools/exp/c4_files.py [skip_oracle=0]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, supersampler_amd as sp
dev = torch.device("cuda", 0)
ctx = sp.Context(0)
out = bench.compare_config4(ctx, dev, len(sys.argv) > 1 and sys.argv[1] == "1", {})
print(json.dumps({k: out[k] for k in ("pipeline_ms", "kernel_ms", "compare_files", "parity_vs_oracle", "parity_sampled_pairs") if k in out}, indent=1))
