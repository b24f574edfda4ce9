"""T2 backward (ring kernel, N = 1e6, K = 16, L = 8, S = 10): launch time with x and Gx as two tensors (10.2 GB read) and with x aliased to
Gx (5.1 GB from HBM, the second stream hits the caches; wrong values, timing only) - how much of the kernel is the x stream?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vmp_for_svae_amd as V
L = V._lib
lib = L.lib()
N, K, Ld, S = 1000000, 16, 8, 10
f32 = dict(dtype=torch.float32, device='cuda')
g = torch.Generator(device='cuda').manual_seed(1)
eta1 = torch.randn(N, Ld, generator=g, **f32); eta2d = -torch.rand(N, Ld, generator=g, **f32) - 0.5
hk = torch.randn(K, Ld, generator=g, **f32); A_ = torch.randn(K, Ld, Ld, generator=g, **f32) * 0.3
Pk = (A_ @ A_.transpose(1, 2) + torch.eye(Ld, **f32)).contiguous()
bias = torch.randn(K, generator=g, **f32); mk = torch.randn(K, Ld, generator=g, **f32); Wk = torch.tril(torch.randn(K, Ld, Ld, generator=g, **f32)).contiguous()
x = torch.randn(N, K, S, Ld, generator=g, **f32); Gx = torch.randn(N, K, S, Ld, generator=g, **f32) * 0.1
lz = torch.log_softmax(torch.randn(N, K, generator=g, **f32), -1); Glz = torch.randn(N, K, generator=g, **f32); GT = torch.randn(N, K, generator=g, **f32)
ge1, ge2 = torch.empty(N, Ld, **f32), torch.empty(N, Ld, **f32)
nblk = lib.vmp_svae_bwd_blocks_for(N, K, Ld, S, 0); PW = lib.vmp_svae_bwd_partial_words(Ld)
part = torch.empty(nblk, K, PW, **f32)
def run(xx):
    L.check(lib.vmp_svae_estep_bwd_n(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk), L.ptr(Wk), None, L.ptr(xx), L.ptr(lz),
                                     L.ptr(Gx), L.ptr(Glz), L.ptr(GT), N, K, Ld, S, L.ptr(ge1), L.ptr(ge2), L.ptr(part), part.numel() * 4, nblk,
                                     L.stream()), 'bwd')
for name, xx in (('x and Gx distinct', x), ('x aliased to Gx', Gx), ('x and Gx distinct', x), ('x aliased to Gx', Gx)):
    for _ in range(3):
        run(xx)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        a.record(); run(xx); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print('%-20s median %.3f ms  min %.3f ms' % (name, ts[len(ts) // 2], ts[0]))
