"""debug: SMM raw moments and one SMM step vs fp64, new vs old library"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
from oracle import mixtures
L = V._lib
for (N, D, K) in ((60, 2, 3), (700, 8, 16), (20000, 8, 16), (7777, 7, 33), (50000, 8, 32)):
    g = torch.Generator(device='cuda').manual_seed(0)
    c = torch.randn(K, D, device='cuda', generator=g) * 5
    x = c[torch.randint(0, K, (N,), device='cuda', generator=g)] + torch.randn(N, D, device='cuda', generator=g)
    r = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
    u = 0.5 + torch.rand(N, K, device='cuda', generator=g)
    st = _mix.raw_stats(x, r, u)
    xd, rd, ud = x.double(), r.double(), u.double()
    w = rd * ud
    ex = torch.cat([rd.sum(0)[:, None], w.sum(0)[:, None], w.t() @ xd, torch.einsum('nk,nd,ne->kde', w, xd, xd).reshape(K, -1)], 1)
    e = (st - ex).abs()
    print('N=%d D=%d K=%d  SMM stats rel err: Nk %.2e Wk %.2e sx %.2e sxx %.2e' % (N, D, K, (e[:, 0].max() / ex[:, 0].abs().max()).item(),
          (e[:, 1].max() / ex[:, 1].abs().max()).item(), (e[:, 2:2 + D].max() / ex[:, 2:2 + D].abs().max()).item(), (e[:, 2 + D:].max() / ex[:, 2 + D:].abs().max()).item()))
    for flav, nm in ((L.VMP_GMM, 'gmm'), (L.VMP_SMM, 'smm')):
        kap = torch.full((K,), 5.0, device='cuda') if flav == L.VMP_SMM else None
        loop = _mix.VMPLoop(x, r, flav, kappa=kap, u_init=u if flav == L.VMP_SMM else None)
        rr = loop.step()
        if flav == L.VMP_SMM:
            ro, uo, _, _ = mixtures.smm_inference_step_chunked(xd.cpu(), rd.cpu(), ud.cpu(), 5.0)
        else:
            ro, _, _, _ = mixtures.gmm_inference_step_chunked(xd.cpu(), rd.cpu())
        print('      %s one step vs fp64 oracle: r max abs err %.2e' % (nm, (rr.double().cpu() - ro).abs().max().item()))
