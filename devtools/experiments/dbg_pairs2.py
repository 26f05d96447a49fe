import sys, torch
sys.path.insert(0, '.')
from lidarcrafter_amd import ops as K
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, Ci, C, H, W, cfg = 8, 64, 64, 32, 1024, 23
x = torch.randn(B, Ci, H, W, device=dev)
w = torch.randn(C, Ci, 3, 3, device=dev) / 17
res = torch.randn(B, C, H, W, device=dev)
gn = K.groupnorm_stats(x, 32, 1e-6)
for rep in range(12):
    y = K.conv2d_ring(x, K.PackedConv(), w, None, tile_cfg=cfg, emit_stats=2, gn_coeffs=gn, res=res)
    e = y._lc_gnstats[(0, C)].buf.double()          # [B, 32, slots, 4]
    # reference per (b, pair, slot): tile 4x64, WPX=4 waves, each wave 2 mfma tiles of 32 px: t = wpx*2+j -> row t//2, col half t%2
    yv = y.double().view(B, C // 2, 2, H // 4, 4, W // 64, 2, 32)      # b, pair, ch, th, row, tw, half, px
    ref = yv.permute(0, 1, 3, 5, 4, 2, 6, 7)                           # b, pair, th, tw, row(=wpx), ch, half, px
    ref_s = ref.reshape(B, C // 2, (H // 4) * (W // 64) * 4, -1)
    rs, rq = ref_s.sum(-1), (ref_s * ref_s).sum(-1)
    p, n, s, q = e[..., 0], e[..., 1], e[..., 2], e[..., 3]
    es, eq = p * n + s, q + 2 * p * s + p * p * n
    bad = ((es - rs).abs() > 1e-3) | ((eq - rq).abs() > 1e-2 * rq.abs().clamp(min=1))
    idx = bad.nonzero()
    print('rep', rep, 'bad entries', len(idx))
    for (b, pr, sl) in idx[:6].tolist():
        print('   b', b, 'pair', pr, '(pair%4 =', pr % 4, ') slot', sl, 'entry', [round(v, 4) for v in e[b, pr, sl].tolist()], 'ref sum', round(float(rs[b, pr, sl]), 4), 'ref sumsq', round(float(rq[b, pr, sl]), 3))
