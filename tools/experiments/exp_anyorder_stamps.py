"""Experiment (needs the -DOMCHAT_EXPERIMENTS=1 library via OMCHAT_LIB): in-kernel clock stamps of the six launches of a batch-1 decode layer (first wave's start,
last wave's end), ordered launches (key 42 = 16) and with the out-of-order o_proj prototype (16 + 7).  The library prints kernel times and boundaries of the last step
when fused_status() is called."""
import os, sys
os.environ["OMCHAT_ALLOW_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from omchat_amd import _lib
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine

lib = _lib.lib()
assert lib.omchat_has_experiments(), "needs OMCHAT_LIB=<experiments twin>"
cfg = omchat13b()
S, STEPS = 3584, 24
# (key 42 value, key 16 value): key 16 = 1 puts the gate|up launch back on the round-3 loop form (one resident round of 512 workgroups)
for val, loop in ((16, 0), (16, 0), (16, 1)):
    lib.omchat_op_set_tuning(42, val); lib.omchat_op_set_tuning(16, loop)
    print(f'key 42 = {val}, key 16 = {loop}', file=sys.stderr, flush=True)
    e = Engine(cfg, dtype="bf16", max_seq=S + STEPS + 8, max_batch=1, max_tiles=1, vision=False)
    e.fill_synthetic(0)
    x = (torch.randn(1, S, cfg.text["hidden_size"], generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16).cuda()
    logits, _ = e.prefill(x, [S]); torch.cuda.synchronize()
    tok = int(torch.argmax(logits[0]))
    for _ in range(STEPS):
        nxt, _ = e.decode_step(torch.tensor([tok])); tok = int(nxt[0])
    torch.cuda.synchronize()
    sys.stderr.flush()
    e.fused_status()
    e.close()
    lib.omchat_op_set_tuning(42, 0); lib.omchat_op_set_tuning(16, 0)
