"""debug: raw moments + one E-step of the T1 kernels vs fp64 torch on the GPU"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
for (N, D, K) in ((60, 2, 3), (700, 8, 16), (4096, 8, 16), (100000, 8, 16), (1000000, 8, 16)):
    g = torch.Generator(device='cuda').manual_seed(0)
    c = torch.randn(K, D, device='cuda', generator=g) * 5
    x = c[torch.randint(0, K, (N,), device='cuda', generator=g)] + torch.randn(N, D, device='cuda', generator=g)
    r = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
    st = _mix.raw_stats(x, r)
    xd, rd = x.double(), r.double()
    ex = torch.cat([rd.sum(0)[:, None], rd.sum(0)[:, None], rd.t() @ xd, torch.einsum('nk,nd,ne->kde', rd, xd, xd).reshape(K, -1)], 1)
    e = (st - ex).abs()
    print('N=%d D=%d K=%d  stats max rel err %.2e  (Nk %.2e, sx %.2e, sxx %.2e)' % (
        N, D, K, (e.max() / ex.abs().max()).item(), (e[:, 0].max() / ex[:, 0].abs().max()).item(),
        (e[:, 2:2 + D].max() / ex[:, 2:2 + D].abs().max()).item(), (e[:, 2 + D:].max() / ex[:, 2 + D:].abs().max()).item()))
    # fused pass: moments of ITS OWN output
    loop = _mix.VMPLoop(x, r, L.VMP_GMM)
    loop.step()
    st2 = loop.stats
    r2 = loop.r.double()
    ex2 = torch.cat([r2.sum(0)[:, None], r2.sum(0)[:, None], r2.t() @ xd, torch.einsum('nk,nd,ne->kde', r2, xd, xd).reshape(K, -1)], 1)
    print('      fused-pass stats vs its own r: max rel err %.2e ; rows sum to 1: %.2e' % (((st2 - ex2).abs().max() / ex2.abs().max()).item(), (r2.sum(1) - 1).abs().max().item()))
    r_e = _mix.estep(x, loop.post['pack'], L.VMP_GMM)[0]
    st3 = _mix.raw_stats(x, loop.r)
    print('      fused r vs E-only r (same pack): max abs diff %.2e ; fused stats vs stand-alone stats of that r: rel %.2e' % (
        (r_e - loop.r).abs().max().item(), ((st2 - st3).abs().max() / st3.abs().max()).item()))
    bad = ((r_e - loop.r).abs().max(1).values > 1e-6).nonzero().flatten()
    if bad.numel():
        print('      rows that differ:', bad[:20].tolist(), '... count', bad.numel())
    from oracle import mixtures, dists
    p = loop.post
    al, be, m, C, v = [p[k].double().cpu() for k in ('alpha', 'beta', 'm', 'C', 'v')]
    r_o = mixtures.gmm_e_step(xd.cpu()[:200000], al, be, m, dists.inv(C), v)[0]
    ef = (loop.r.double().cpu()[:200000] - r_o).abs().max(1).values
    ee = (r_e.double().cpu()[:200000] - r_o).abs().max(1).values
    print('      vs fp64 oracle (posterior as stored, fp32): fused max %.2e (rows > 1e-5: %s)  E-only max %.2e (rows > 1e-5: %s)' % (
        ef.max().item(), (ef > 1e-5).nonzero().flatten()[:10].tolist(), ee.max().item(), (ee > 1e-5).nonzero().flatten()[:10].tolist()))
    if bad.numel():
        torch.set_printoptions(precision=5, linewidth=220, sci_mode=False)
        for b in bad[:3].tolist():
            print('      row', b, 'fused :', loop.r[b].cpu())
            print('      row', b, 'E-only:', r_e[b].cpu())
            dd = (loop.r[b] - r_e[b]).cpu()
            print('      diff:', dd, 'sum', dd.sum().item())
            print('      x   :', x[b].cpu())
