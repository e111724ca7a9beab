"""ViT + projector throughput vs tile batch (not product): shows the tile-quantisation effect of M = 1025 * B."""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import synth
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine
cfg = omchat13b()
for B in (1, 3, 6, 12, 24):
    e = Engine(cfg, dtype="bf16", max_tiles=B, text=False)
    e.fill_synthetic(0)
    px = torch.randn(B, 3, 448, 448, device="cuda").bfloat16()
    for _ in range(2): e.encode_images(px)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 3
    e0.record()
    for _ in range(it): e.encode_images(px)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    tf = B * (11.945 + 0.0498) / (ms / 1e3)
    print(f"B={B:3d}: {ms:8.2f} ms  {B/(ms/1e3):7.1f} tiles/s  {tf:7.1f} TFLOP/s = {tf/2500*100:5.1f} % of 2.5 PF", flush=True)
    e.close(); del e
    torch.cuda.empty_cache()
