import sys, os
sys.path.insert(0, '/root/repo')
import torch
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer
from torch.profiler import profile, ProfilerActivity
vae.reset_variables()
tr = SVAETrainer(10, 8, 50, 6, nb_samples=10)
y = torch.randn(64, 6, device='cuda')
for _ in range(3): out = tr.step(y)
names, params = tr.trainables()
grads = [out['grads'][n] for n in names]
for n, p, g in zip(names, params, grads):
    print('%-40s p %s %s  g %s %s contiguous %s offset %d' % (n, tuple(p.shape), p.stride(), tuple(g.shape), g.stride(), g.is_contiguous(), g.storage_offset()))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.opt.apply_gradients(grads)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=14, max_name_column_width=50))
