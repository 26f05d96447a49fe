# phases of the level-0 conv at batch 1 (one tile per block): where do 24 us go?
for s in 1:64:64:32:1024 1:128:128:16:512 8:64:64:32:1024; do
python devtools/conv_phases.py $s --gn --emit
done
python - <<'PY'
import torch, time
import sys; sys.path.insert(0, '.')
from lidarcrafter_amd import ops as K
dev = torch.device('cuda:0')
for (B,Ci,Co,H,W) in ((1,64,64,32,1024),(1,128,128,16,512),(1,256,256,8,256),(1,512,512,4,128),(2,64,64,32,1024)):
    x = torch.randn(B,Ci,H,W,device=dev); w = torch.randn(Co,Ci,3,3,device=dev)/(Ci*9)**.5; b = torch.randn(Co,device=dev)
    pk = K.PackedConv(); out = torch.empty(B,Co,H,W,device=dev)
    gn = K.groupnorm_stats(x, 8, 1e-6)
    for cfg in (0, 13, 25, 23):
        try:
            f = lambda: K.conv2d_ring(x, pk, w, b, out=out, precision='f16x2', gn_coeffs=gn, emit_stats=True, tile_cfg=cfg)
            for _ in range(3): f()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20): f()
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"{B}:{Ci}:{Co}:{H}:{W} cfg {cfg}: {dt*1e6:.1f} us")
        except Exception as e:
            print(f"{B}:{Ci}:{Co}:{H}:{W} cfg {cfg}: {type(e).__name__} {str(e)[:80]}")
PY
