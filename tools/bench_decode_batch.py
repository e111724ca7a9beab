"""Decode step time of the full 13B decoder vs batch size (not product): where the batched (packed MFMA) path beats b x the batch-1 path."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine
cfg = omchat13b()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
for b in (1, 2, 3, 4, 8, 16, 32):
    eng = Engine(cfg, dtype="bf16", max_seq=S + 160, max_batch=b, vision=False)
    eng.fill_synthetic(0)
    x = (torch.randn(b, S, 3584, device="cuda") * 0.5).bfloat16()
    logits, _ = eng.prefill(x)
    tok = eng.argmax(logits)
    for _ in range(8): tok, _ = eng.decode_step(tok)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 64
    for _ in range(n): tok, _ = eng.decode_step(tok)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"b={b:2d}: {dt*1e3:.3f} ms/step  {b/dt:8.1f} tok/s", flush=True)
    eng.close(); del eng, x
    torch.cuda.empty_cache()
