import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench, supersampler_amd as sp
dev = torch.device("cuda", 0)
ctx = sp.Context(0)
out = bench.scan_config5(ctx, dev, False, {})
print(json.dumps({k: out[k] for k in ("scan_pipeline_ms", "sketch_keys", "parity_vs_oracle")}, indent=1))
