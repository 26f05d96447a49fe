import sys, torch
sys.path.insert(0, '.')
from lidarcrafter_amd import ops as K
dev = torch.device('cuda:0')
torch.manual_seed(0)
bad = 0
for rep in range(2):
  for (B, Ci, C, H, W, cfg) in ((1, 64, 64, 64, 2048, 0), (1, 34, 64, 64, 2048, 0), (1, 128, 64, 64, 2048, 0), (2, 64, 64, 8, 128, 23), (8, 64, 64, 32, 1024, 23), (2, 32, 64, 8, 128, 25), (2, 32, 64, 8, 128, 22), (2, 32, 128, 8, 128, 13)):
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(C, Ci, 3, 3, device=dev) / 17
    res = torch.randn(B, C, H, W, device=dev)
    gn = K.groupnorm_stats(x, 2 if Ci == 34 else 32, 1e-6)
    for unit in (2, True):
      for kw in ({}, {"res": res}, {"gn_coeffs": gn, "res": res}):
        y = K.conv2d_ring(x, K.PackedConv(), w, None, tile_cfg=cfg, emit_stats=unit, **kw)
        h = y._lc_gnstats[(0, C)]
        e = h.buf
        u = h.unit
        p, n, s, q = e[..., 0].double(), e[..., 1].double(), e[..., 2].double(), e[..., 3].double()
        N = n.sum(2)
        mean = ((p * n + s).sum(2) / N)
        ex2 = ((q + 2 * p * s + p * p * n).sum(2) / N)
        yv = y.view(B, C // u, u * H * W).double()
        err = float((mean - yv.mean(2)).abs().max())
        err2 = float((ex2 - (yv * yv).mean(2)).abs().max() / (yv * yv).mean())
        nn = int(torch.isnan(e).sum())
        if nn or not err < 1e-6 or not err2 < 1e-5 or float((N - u * H * W).abs().max()) != 0:
            bad += 1
            print((B, Ci, C, H, W, cfg), sorted(kw), 'unit', u, 'nan', nn, 'mean err', err, 'ex2 err', err2, 'N', float(N.min()), float(N.max()))
print('bad cases', bad)
