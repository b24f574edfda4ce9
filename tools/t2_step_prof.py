"""T2 step (bench.bench_t2) as a profiling target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.bench_t2(1_000_000, 8, 16, 10, 5, 2, torch.device('cuda', 0))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('ms_per_step', 'fwd_kernel_ms', 'bwd_kernel_ms')})
