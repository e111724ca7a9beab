"""debug: which phase of the one-launch layer differs from the six launches (tiny geometry, 1 layer); zeroing weights isolates phases"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from omchat_amd import synth, _lib
from omchat_amd.config import tiny
from omchat_amd.engine import Engine
from gpu_util import rnd

lib = _lib.lib()
def run(cfg, sd, k23, dt="bf16", S=21, steps=2):
    _lib.check(lib.omchat_op_set_tuning(23, k23)); _lib.check(lib.omchat_op_set_tuning(22, 0))
    e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, max_tiles=1, vision=False)
    e.load_state_dict(sd)
    x = rnd(torch.randn(1, S, cfg.text["hidden_size"], generator=torch.Generator().manual_seed(3)) * 0.5, dt)
    e.prefill(x)
    outs = []
    tok = torch.tensor([11])
    for _ in range(steps):
        nxt, lg = e.decode_step(tok, want_logits=True)
        outs.append(lg.float().cpu().clone()); tok = nxt.cpu()
    torch.cuda.synchronize()
    print("   fused status", e.fused_status())
    e.close()
    return outs

for name, kw in (("q4kv2", dict(q_heads=4, kv_heads=2, layers_t=1)), ("q7kv1", dict(q_heads=7, kv_heads=1, layers_t=1))):
    cfg = tiny(**kw)
    base = {k: v for k, v in synth.state_dict(cfg, 5).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    for variant in ("no_mlp(down=0)",):
        sd = dict(base)
        if variant.startswith("no_mlp"):
            sd["model.layers.0.mlp.down_proj.weight"] = np.zeros_like(base["model.layers.0.mlp.down_proj.weight"])
        if variant.startswith("no_attn"):
            sd["model.layers.0.self_attn.o_proj.weight"] = np.zeros_like(base["model.layers.0.self_attn.o_proj.weight"])
        if variant != "full":
            sd["model.layers.0.mlp.down_proj.weight"] = np.zeros_like(base["model.layers.0.mlp.down_proj.weight"])
        for pz in ("q", "k", "v"):
            if variant.startswith(pz + "=0"):
                sd[f"model.layers.0.self_attn.{pz}_proj.weight"] = np.zeros_like(base[f"model.layers.0.self_attn.{pz}_proj.weight"])
                sd[f"model.layers.0.self_attn.{pz}_proj.bias"] = np.zeros_like(base[f"model.layers.0.self_attn.{pz}_proj.bias"])
        if variant == "no_bias":
            for p in ("q", "k", "v"):
                sd[f"model.layers.0.self_attn.{p}_proj.bias"] = np.zeros_like(base[f"model.layers.0.self_attn.{p}_proj.bias"])
        a = run(cfg, sd, 1); b = run(cfg, sd, 0)
        for s_, (x_, y_) in enumerate(zip(a, b)):
            d = (x_ - y_).abs()
            print(name, variant, "step", s_, "equal", bool(torch.equal(x_, y_)), "max abs diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel())
_lib.check(lib.omchat_op_set_tuning(23, 1))
