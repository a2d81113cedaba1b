import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, supersampler_amd as sp
dev = torch.device("cuda", 0)
ctx = sp.Context(0)
out = bench.scan_config5(ctx, dev, os.environ.get("C5_SKIP_ORACLE") == "1", {}, gbp=float(sys.argv[1]) if len(sys.argv) > 1 else 4.0)
print(json.dumps({k: out.get(k) for k in ("scan_pipeline_ms", "dense_kernel_ms", "packed_2bit", "sketch_keys", "sketch_file", "parity_vs_oracle")}, indent=1))
