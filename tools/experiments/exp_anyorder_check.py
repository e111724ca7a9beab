"""Experiment (not product; needs the -DOMCHAT_EXPERIMENTS=1 library via OMCHAT_LIB): are the decode step's results still right under an out-of-order launch
key?  python tools/exp_anyorder_check.py [key]: 41 (default) = every launch goes out with hipExtAnyOrderLaunch (no dependencies: expected WRONG), 42 = only the
batch-1 o_proj, behind the split-KV merge, waiting on the merge's completion flags (expected bit-identical).  Qwen2-7B widths, 28 layers, synthetic weights: prefill of 600 positions, then 48 greedy decode steps with their
logits; key 41 = 0 twice (run-to-run determinism), key 41 = 1 three times; every step's logits compared bit for bit with the first run."""
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
S, STEPS = 600, 48


KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 41
VAL = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def run(key):
    lib.omchat_op_set_tuning(KEY, key)
    e = Engine(cfg, dtype="bf16", max_seq=S + STEPS + 8, max_batch=1, max_tiles=1, vision=False)
    e.fill_synthetic(0)
    x = (torch.randn(1, S, cfg.text["hidden_size"], generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16).cuda()
    logits, _ = e.prefill(x, [S]); torch.cuda.synchronize()
    out = [logits[0].float().cpu()]
    tok = int(torch.argmax(out[0]))
    ids = [tok]
    for _ in range(STEPS):
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True)
        out.append(lg[0].float().cpu()); tok = int(nxt[0]); ids.append(tok)
    torch.cuda.synchronize()
    bits = e.fused_status()[1]
    e.close()
    lib.omchat_op_set_tuning(KEY, 0)
    if bits:
        print(f'  time-out bits of the in-kernel waits: {bits:#x}')
    return out, ids


base, ids0 = run(0)
for name, key in (("in-order again", 0), (f"key {KEY} = {VAL}, run 1", VAL), (f"key {KEY} = {VAL}, run 2", VAL), (f"key {KEY} = {VAL}, run 3", VAL)):
    o, ids = run(key)
    bad = [k for k in range(len(base)) if not torch.equal(o[k], base[k])]
    worst = max((float((o[k] - base[k]).norm() / base[k].norm()) for k in bad), default=0.0)
    print(f"{name}: {len(bad)} of {len(base)} logit rows differ from the first in-order run (first at step {bad[0] if bad else '-'}, worst rel {worst:.3e}); "
          f"ids equal: {ids == ids0}", flush=True)
