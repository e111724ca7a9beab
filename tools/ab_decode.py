"""A/B of decode-step options on the full 13B geometry (not product): tokens/s with a tuning key on and off."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine
key = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = omchat13b()
eng = Engine(cfg, dtype="bf16", max_seq=4096, max_batch=1, vision=False)
eng.fill_synthetic(0)
x = (torch.randn(1, 3584, 3584, device="cuda") * 0.5).bfloat16()
for fp8 in (0, 1):
    eng.enable_fp8_decode(bool(fp8))
    for val in (1, 0, 1, 0):
        _lib.lib().omchat_op_set_tuning(key, val)
        logits, _ = eng.prefill(x)
        tok = eng.argmax(logits)
        for _ in range(8): tok, _ = eng.decode_step(tok)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks = []
        for _ in range(128):
            tok, _ = eng.decode_step(tok); toks.append(tok)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 128
        print(f"fp8={fp8} key{key}={val}: {dt*1e3:.3f} ms/token  {1/dt:.1f} tok/s  ids {[int(t) for t in toks[:4]]}", flush=True)
_lib.lib().omchat_op_set_tuning(key, 1)
