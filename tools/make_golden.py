#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (om-ai-lab/OmChat mounted at /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Imports the reference's
Python with stub modules for the absent `timm` / `peft` (SURVEY.md §8c recipe), instantiates its own
classes at tiny dimensions, loads the deterministic synthetic weights of `omchat_amd.synth` into them
(key names are the reference's own), runs them on CPU and dumps inputs + outputs as small .npz files under
tests/golden/.  Nothing from the reference's source is copied: fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import os, sys, types, importlib.machinery
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    import transformers  # noqa: F401  (must be imported before the stubs)
    from transformers import Qwen2Config  # noqa: F401

    class DropPath(torch.nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    _stub("timm"); _stub("timm.models"); _stub("timm.models.layers", DropPath=DropPath)
    _stub("timm.layers", LayerNorm=torch.nn.LayerNorm, LayerNorm2d=torch.nn.LayerNorm)
    _stub("timm.models.regnet", RegStage=object)
    _stub("peft", PeftModel=object)
    sys.path.insert(0, REF)
    import omchat.model  # noqa: F401
    return sys.modules["omchat"]


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    import_reference()
    from omchat_amd import synth
    from omchat_amd.config import tiny, OmChatConfig
    from omchat.model.multimodal_encoder.intern_vit_6b.modeling_intern_vit import (
        InternVisionModel, InternAttention, InternRMSNorm)
    from omchat.model.multimodal_encoder.intern_vit_6b.configuration_intern_vit import InternVisionConfig
    from omchat.model.multimodal_projector.builder import build_vision_projector
    import omchat.model.multimodal_encoder.internVIT_encoder as enc_mod
    from omchat.model.language_model.omchat_qwen2 import OmChatQwen2Config, OmChatQwen2ForCausalLM
    from omchat.mm_utils import select_best_resolution, tokenizer_image_token
    os.makedirs(OUT, exist_ok=True)
    TOW = synth.TOWER

    # ------------------------------------------------------------------ a2: RMSNorm, production width
    x = T(synth.uniform("g.rms.x", (8, 3200), 0, 1.0))
    wn = T(synth.uniform("g.rms.w", (3200,), 0, 0.05, 1.0))
    m = InternRMSNorm(3200, eps=1e-6); m.weight.data.copy_(wn)
    save("rmsnorm_3200", x=x, w=wn, y=m(x), y_half=m(x.half()).float())

    # ------------------------------------------------------------------ a1,a3,a5,a6,a7: tiny tower (fp32)
    cfg = tiny()
    vc = InternVisionConfig(**{**cfg.vision, "use_flash_attn": False})
    tower = InternVisionModel(vc).eval()
    sd = synth.state_dict(cfg, seed=0, only_prefix=TOW)
    missing = tower.load_state_dict({k[len(TOW):]: T(v) for k, v in sd.items()}, strict=True)
    px = T(synth.pixels(2, cfg.vision["image_size"], seed=3))
    out = tower(px, output_hidden_states=True, return_dict=True)
    lay0 = tower.encoder.layers[0]
    h0 = out.hidden_states[0]
    n1 = lay0.norm1(h0)
    save("vit_tiny", pixels=px, seed=0, pixel_seed=3,
         hs0=out.hidden_states[0], hs1=out.hidden_states[1], hs2=out.hidden_states[2],
         l0_norm1=n1, l0_attn=lay0.attn(n1), l0_mlp=lay0.mlp(lay0.norm2(out.hidden_states[0])))

    # tower wrapper (reference casts pixels to fp16 unconditionally, internVIT_encoder.py:53) -> fp16 weights
    orig_cfg = enc_mod.InternVisionConfig
    enc_mod.InternVisionConfig = lambda *a, **k: vc
    try:
        for sel_layer, sel_feat in ((-1, "patch"), (-2, "patch"), (-1, "cls_patch")):
            args = types.SimpleNamespace(mm_vision_select_layer=sel_layer, mm_vision_select_feature=sel_feat)
            tw = enc_mod.InternVITVisionTower("internvit-6b-448px", args, delay_load=False)
            tw.vision_tower.load_state_dict({k[len(TOW):]: T(v) for k, v in sd.items()}, strict=True)
            tw.vision_tower.half()
            feats = tw(px.half())
            save(f"tower_wrapper_L{sel_layer}_{sel_feat}", pixels=px, feats_half=feats.float(),
                 select_layer=sel_layer, select_feature=sel_feat)
    finally:
        enc_mod.InternVisionConfig = orig_cfg

    # ------------------------------------------------------------------ a1: pos-embed bicubic resize path
    cfg_r = tiny(image_size=112)
    vcr = InternVisionConfig(**{**cfg_r.vision, "use_flash_attn": False})
    tr = InternVisionModel(vcr).eval()
    sdr = synth.state_dict(cfg_r, seed=5, only_prefix=TOW)
    tr.load_state_dict({k[len(TOW):]: T(v) for k, v in sdr.items()}, strict=True)
    pxr = T(synth.pixels(1, 56, seed=4))
    save("vit_embed_resize", pixels=pxr, emb=tr.embeddings(pxr), seed=5, pixel_seed=4)

    # ------------------------------------------------------------------ a3: full-width attention (25 heads x 128), N=33
    vfull = InternVisionConfig(use_flash_attn=False)           # reference defaults: 3200 / 25 heads
    att = InternAttention(vfull).eval()
    P = "g.attnfull."
    wq = {"qkv.weight": (9600, 3200), "q_norm.weight": (3200,), "k_norm.weight": (3200,),
          "proj.weight": (3200, 3200), "proj.bias": (3200,)}
    st = {k: T(synth.uniform(P + k, s, 0, 0.05 if "norm" in k else 0.02, 1.0 if "norm" in k else 0.0)) for k, s in wq.items()}
    att.load_state_dict(st, strict=True)
    xa = T(synth.uniform(P + "x", (1, 33, 3200), 0, 1.0))
    save("vit_attn_full", x=xa, y=att(xa))

    # ------------------------------------------------------------------ a9: projector
    pc = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=cfg.vision["hidden_size"],
                               hidden_size=cfg.text["hidden_size"])
    proj = build_vision_projector(pc).eval()
    sdp = synth.state_dict(cfg, seed=0, only_prefix="model.mm_projector.")
    proj.load_state_dict({k[len("model.mm_projector."):]: T(v) for k, v in sdp.items()}, strict=True)
    xp = T(synth.uniform("g.proj.x", (2, 16, cfg.vision["hidden_size"]), 0, 1.0))
    save("projector_tiny", x=xp, y=proj(xp))

    # ------------------------------------------------------------------ whole tiny model (splice, decoder, e2e)
    def build_model(c, seed, dtype=torch.float32):
        vcc = InternVisionConfig(**{**c.vision, "use_flash_attn": False})
        enc_mod.InternVisionConfig = lambda *a, **k: vcc
        try:
            qc = OmChatQwen2Config(
                hidden_size=c.text["hidden_size"], intermediate_size=c.text["intermediate_size"],
                num_hidden_layers=c.text["num_hidden_layers"], num_attention_heads=c.text["num_attention_heads"],
                num_key_value_heads=c.text["num_key_value_heads"], vocab_size=c.text["vocab_size"],
                head_dim=c.text["head_dim"], rms_norm_eps=1e-6, rope_theta=1e6, max_position_embeddings=4096,
                tie_word_embeddings=False, attn_implementation="eager",
                mm_vision_tower="internvit-6b-448px", mm_projector_type="mlp2x_gelu",
                mm_hidden_size=c.vision["hidden_size"], mm_vision_select_layer=-1, delay_load=False)
            try:
                qc.rope_parameters = {"rope_type": "default", "rope_theta": 1e6}
            except Exception:
                pass
            model = OmChatQwen2ForCausalLM(qc).eval()
        finally:
            enc_mod.InternVisionConfig = orig_cfg
        full = synth.state_dict(c, seed=seed)
        res = model.load_state_dict({k: T(v) for k, v in full.items()}, strict=False)
        bad = [k for k in res.missing_keys if "inv_freq" not in k]
        assert not bad and not res.unexpected_keys, (bad, res.unexpected_keys)
        model.config._attn_implementation = "eager"
        return model.to(dtype), full

    model, full = build_model(cfg, 0)
    assert model.config._attn_implementation == "eager"
    H = cfg.text["hidden_size"]
    ntok = cfg.num_image_tokens        # 16
    emb = full["model.embed_tokens.weight"]

    # ---- a11 splice cases (encode_images replaced by given features so the fixture isolates the splice)
    def run_splice(name, ids, mask, n_tiles, side="right", maxlen=None):
        feats = T(synth.uniform("g.splice.feats." + name, (n_tiles, ntok, H), 0, 1.0))
        model.encode_images = lambda images: feats
        model.config.tokenizer_padding_side = side
        model.config.tokenizer_model_max_length = maxlen
        ids_t = torch.tensor(ids, dtype=torch.long)
        mask_t = None if mask is None else torch.tensor(mask, dtype=torch.long)
        dummy = torch.zeros(n_tiles, 3, 56, 56)
        r = model.prepare_inputs_labels_for_multimodal(ids_t, None, mask_t, None, None, dummy)
        assert r[0] is None and r[1] is None
        save("splice_" + name, ids=ids_t, mask=(np.zeros(0) if mask is None else mask_t), has_mask=mask is not None,
             feats=feats, embeds=r[4], mask_out=(np.zeros(0) if r[2] is None else r[2]), side=side,
             maxlen=-1 if maxlen is None else maxlen, seed=0)
        del model.encode_images

    I = -200
    run_splice("1x3", [[5, I, 7, 8, I, 9, I, 10, 11]], None, 3)
    run_splice("2_uneven_right", [[1, 2, I, 3, 4, I, 5, 6, 7, 8], [9, I, 10, 11, 0, 0, 0, 0, 0, 0]],
               [[1] * 10, [1] * 4 + [0] * 6], 3, "right")
    run_splice("2_uneven_left", [[1, 2, I, 3, 4, I, 5, 6, 7, 8], [9, I, 10, 11, 0, 0, 0, 0, 0, 0]],
               [[1] * 10, [1] * 4 + [0] * 6], 3, "left")
    run_splice("noimage_row", [[1, 2, I, 3], [4, 5, 6, 7]], [[1] * 4, [1] * 4], 2)
    run_splice("truncate", [[5, I, 7, 8, I, 9]], [[1] * 6], 2, "right", 24)
    model.config.tokenizer_padding_side = "right"; model.config.tokenizer_model_max_length = None

    # decode short-circuit (omchat_arch.py:61-70): legacy tuple cache probe
    am = torch.ones(2, 5, dtype=torch.long); am[1, :2] = 0
    legacy = ((torch.zeros(2, 1, 9, 4), torch.zeros(2, 1, 9, 4)),)
    r = model.prepare_inputs_labels_for_multimodal(torch.zeros(2, 1, dtype=torch.long), None, am, legacy, None, torch.zeros(1, 3, 56, 56))
    save("splice_decode_shortcircuit", mask_in=am, past_len=9, mask_out=r[2], position_ids=r[1])

    # ---- a13-a18: decoder prefill (inputs_embeds path) + 4 manual decode steps, eager attention, fp32
    def run_decoder(name, c, seed, S=12, steps=4):
        mdl, fl = build_model(c, seed)
        x = T(synth.uniform("g.dec.x." + name, (1, S, c.text["hidden_size"]), 0, 1.0))
        o = mdl(inputs_embeds=x, use_cache=True)
        logits = [o.logits[0]]
        cache = o.past_key_values
        toks = []
        for s in range(steps):
            nxt = int(torch.argmax(logits[-1][-1].float()))
            toks.append(nxt)
            o = mdl(input_ids=torch.tensor([[nxt]]), past_key_values=cache, use_cache=True)
            cache = o.past_key_values
            logits.append(o.logits[0])
        # hidden state after layer 0 for finer-grained checks
        hs = mdl.model(inputs_embeds=x, output_hidden_states=True).hidden_states
        save("decoder_" + name, x=x, seed=seed, prefill_logits=logits[0], step_logits=torch.stack([l[0] for l in logits[1:]]),
             tokens=toks, hs1=hs[1][0], q_heads=c.text["num_attention_heads"], kv_heads=c.text["num_key_value_heads"])

    run_decoder("7q1kv", tiny(q_heads=7, kv_heads=1), 11)
    run_decoder("4q2kv", tiny(q_heads=4, kv_heads=2), 12)

    # ---- a10-a19 end-to-end: fp16-on-CPU reference plumbing, manual greedy loop
    def run_e2e(name, c, seed, ids, n_tiles, new_tokens):
        mdl, fl = build_model(c, seed, torch.float16)
        px16 = T(synth.pixels(n_tiles, c.vision["image_size"], seed=7)).half()
        ids_t = torch.tensor([ids], dtype=torch.long)
        o = mdl(input_ids=ids_t, images=px16, use_cache=True)
        prefill_logits = o.logits[0].float()
        cache = o.past_key_values
        toks, margins = [], []
        last = o.logits[0, -1].float()
        for s in range(new_tokens):
            top2 = torch.topk(last, 2)
            nxt = int(torch.argmax(last)); toks.append(nxt); margins.append(float(top2.values[0] - top2.values[1]))
            o = mdl(input_ids=torch.tensor([[nxt]]), past_key_values=cache, use_cache=True)   # images=None -> :61,:70 early return
            cache = o.past_key_values
            last = o.logits[0, -1].float()
        # fp32 run of the same model through the same entry point is impossible (fp16 cast, internVIT_encoder.py:53);
        # record the image features in fp16 for the encode_images check
        feats = mdl.encode_images(px16).float()
        save("e2e_" + name, ids=ids_t, n_tiles=n_tiles, pixel_seed=7, seed=seed, prefill_logits_last=prefill_logits[-1],
             prefill_len=prefill_logits.shape[0], tokens=toks, margins=margins, image_features_half=feats)

    run_e2e("tiny", tiny(), 21, [3, I, 17, 18, I, 19, 20, 21, 5, 9], 2, 16)

    # ---- pure-int tables: anyres tiling + image-token tokenisation layout (mm_utils.py:12-39,197-230)
    pin = [[448, 896], [896, 448], [896, 896], [1344, 448], [448, 1344], [1344, 1344]]
    sizes = [(448, 448), (570, 380), (1000, 667), (1344, 448), (300, 900), (2000, 1000), (100, 100)]
    best = [select_best_resolution(s, pin) for s in sizes]

    class Tok:                       # stub tokenizer: one id per character, no BOS (Qwen2 has none)
        bos_token_id = None
        def __call__(self, s):
            return types.SimpleNamespace(input_ids=[1000 + ord(ch) for ch in s])
    prompt = "<image>\npatch:<image>\npatch:<image>\nhello"
    lay = tokenizer_image_token(prompt, Tok(), -200)
    save("int_tables", sizes=np.array(sizes), best=np.array(best), prompt_ids=np.array(lay))
    print("done")


if __name__ == "__main__":
    main()
