"""Achievable HBM rates of pure-write, pure-read and copy streams on this box (torch kernels, 5.12 GB = the (N,K,S,L) sample
tensor of C3): the yardstick for the write-dominated E-step forward with in-kernel noise."""
import torch
n = 1_280_000_000
x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
gb = n * 4 / 1e9
for name, f, by in (('fill_ (write)', lambda: x.fill_(1.0), gb), ('zero_ (memset)', lambda: x.zero_(), gb), ('sum (read)', lambda: x.sum(), gb),
                    ('copy_ (read+write)', lambda: y.copy_(x), 2 * gb), ('mul_ in place (read+write)', lambda: x.mul_(1.0001), 2 * gb),
                    ('normal_ (write, Philox)', lambda: x.normal_(), gb)):
    ms = t(f)
    print('%-28s %.3f ms  %.0f GB/s' % (name, ms, by / ms * 1e3))
