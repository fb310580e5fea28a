import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from pistoseg_amd import ops
D = torch.device('cuda:0')
def planes(t, weights=False):
    hi = t.to(torch.bfloat16); lo = (t - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, hi, lo] if weights else [hi, lo, hi], dim=-1).contiguous()
for (n, h, w, cin, cout) in ((33, 28, 28, 704, 1024), (33, 28, 28, 704, 2048), (40, 28, 28, 704, 1024)):
    spec = ops.ConvSpec(cin, cout, 1, 1, 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g); wt = torch.randn(cout, 1, 1, cin, generator=g) * 0.05
    xd = planes(x).to(D); wf = planes(wt, True).to(D)
    out = torch.full((n, h, w, 3 * cout), float('nan'), device=D, dtype=torch.bfloat16)
    ops.conv2d_fwd(spec, xd, wf, out_raw=out, split=True)
    torch.cuda.synchronize()
    o = out.view(-1, 3 * cout).float().cpu()
    p0, p1, p2 = o[:, :cout], o[:, cout:2 * cout], o[:, 2 * cout:]
    bad = ~(p0 == p2)
    print((n, cin, cout), 'nan p0', int(p0.isnan().sum()), 'nan p1', int(p1.isnan().sum()), 'nan p2', int(p2.isnan().sum()), 'mismatch', int(bad.sum()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print(' rows', rows[:10].tolist(), '...', rows[-5:].tolist(), len(rows), ' cols', cols[:10].tolist(), '...', cols[-5:].tolist(), len(cols))
        r, c = bad.nonzero()[0].tolist(); print(' first', r, c, p0[r, c].item(), p2[r, c].item())
