#!/usr/bin/env python3
"""Round-2 golden vectors from the REAL reference (same recipe and rules as tools/make_golden.py: runs only in the build
container, the fixtures are data only):

  make_context.npz / .json   context_tokens + raw_text of omchat/make_context.py:66-148 (history window, system prompt, image
                             sentinels) and the query layout of get_context (:14-43) for a stub one-id-per-character tokenizer
  leftpad_prefill.npz        OmChatQwen2ForCausalLM.forward on a LEFT-padded batch of two samples with uneven tiles
                             (omchat_arch.py:176-184, position_ids dropped :206-207 -> arange(S)): logits of the last position,
                             next to the right-padded run of the same batch
  e2e_free_f16.npz           a free-running fp16 greedy sequence whose every top-1 / top-2 margin is far above fp16 noise:
                             the HIP path must reproduce every id without teacher forcing

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_r2.py
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import import_reference, T, save, OUT      # noqa: E402


class Tok:
    """stub tokenizer: one id per character (1000 + ord), no BOS (Qwen2 has none)"""
    bos_token_id = None

    def encode(self, s):
        return [1000 + ord(ch) for ch in s]

    def __call__(self, s):
        return types.SimpleNamespace(input_ids=self.encode(s))


def main():
    only_free = "--only-free" in sys.argv
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    import_reference()
    from omchat_amd import synth
    from omchat_amd.config import tiny
    from omchat.make_context import make_context, get_context
    import omchat.make_context as mc_mod
    import omchat.model.multimodal_encoder.internVIT_encoder as enc_mod
    from omchat.model.multimodal_encoder.intern_vit_6b.configuration_intern_vit import InternVisionConfig
    from omchat.model.language_model.omchat_qwen2 import OmChatQwen2Config, OmChatQwen2ForCausalLM

    # ------------------------------------------------------------------ make_context / get_context
    if only_free:
        return main_models(synth, tiny, enc_mod, InternVisionConfig, OmChatQwen2Config, OmChatQwen2ForCausalLM, True)
    tok = Tok()
    cases = []
    hist = [("first question", "first answer"), ("<image>\nwhat is this", "a cat"), ("third", "ok")]
    for name, query, history, system, window in (
            ("plain", "hello there", None, "You are a helpful assistant.", 6144),
            ("image3", "<image>\npatch:<image>\npatch:<image>\ndescribe", None, "You are a helpful assistant.", 6144),
            ("history_all", "and now?", hist, "sys", 6144),
            ("history_window", "and now?", hist, "sys", 120),       # only the newest turns fit (make_context.py:118-126)
            ("history_none_fits", "q", hist, "sys", 10),
            ("empty_system", "x", None, "", 6144)):
        raw, ids = make_context(tok, query, history, system, window)
        cases.append(dict(name=name, query=query, history=history, system=system, max_window_size=window, raw_text=raw, context_tokens=ids))
    raw, ids = make_context(tok, "raw prompt", None, "", 6144, "raw")
    cases.append(dict(name="raw", query="raw prompt", history=None, system="", max_window_size=6144, chat_format="raw", raw_text=raw, context_tokens=ids))
    # get_context (:14-43) with an image: tiles come from process_anyres_image; stub it to n tiles, and let .cuda() be a no-op on CPU
    orig_pai, orig_cuda = mc_mod.process_anyres_image, torch.Tensor.cuda
    try:
        torch.Tensor.cuda = lambda self, *a, **k: self
        for n in (1, 3, 5):
            mc_mod.process_anyres_image = lambda image, ip, pins, flag, return_best_res=False, n=n: ([torch.zeros(3, 4, 4)] * n, (448, 896))
            inp, ids, image_tensor = get_context("what is <image> shown here ", tok, image=object(), image_processor=None, image_grid_pinpoints=None)
            cases.append(dict(name=f"get_context_{n}", text="what is <image> shown here ", n_tiles=n, raw_text=inp, context_tokens=ids,
                              image_tensor_shape=list(image_tensor.shape), image_tensor_dtype=str(image_tensor.dtype)))
    finally:
        mc_mod.process_anyres_image, torch.Tensor.cuda = orig_pai, orig_cuda
    with open(os.path.join(OUT, "make_context.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print(f"  wrote make_context.json ({len(cases)} cases)")

    main_models(synth, tiny, enc_mod, InternVisionConfig, OmChatQwen2Config, OmChatQwen2ForCausalLM, False)


def main_models(synth, tiny, enc_mod, InternVisionConfig, OmChatQwen2Config, OmChatQwen2ForCausalLM, only_free):
    # ------------------------------------------------------------------ tiny whole model
    orig_cfg = enc_mod.InternVisionConfig

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

    # ------------------------------------------------------------------ left-padded batch prefill (fp32; image features given)
    cfg = tiny()
    I = -200
    if not only_free:
        leftpad(cfg, build_model, synth)
    free_running(cfg, build_model, synth)
    print("done")


def leftpad(cfg, build_model, synth):
    model, _ = build_model(cfg, 31)
    H, ntok = cfg.text["hidden_size"], cfg.num_image_tokens
    I = -200
    ids = torch.tensor([[1, 2, I, 3, 4, I, 5, 6, 7, 8], [9, I, 10, 11, 0, 0, 0, 0, 0, 0]], dtype=torch.long)
    mask = torch.tensor([[1] * 10, [1] * 4 + [0] * 6], dtype=torch.long)
    feats = T(synth.uniform("g.leftpad.feats", (3, ntok, H), 0, 1.0))
    model.encode_images = lambda images: feats
    dummy = torch.zeros(3, 3, 56, 56)
    res = {}
    for side in ("left", "right"):
        model.config.tokenizer_padding_side = side
        model.config.tokenizer_model_max_length = None
        o = model(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True)
        res[side] = o.logits
    S = res["left"].shape[1]
    lens = [8 + 2 * ntok, 3 + ntok]
    save("leftpad_prefill", ids=ids, mask=mask, feats=feats, seed=31, S=S, lengths=np.array(lens),
         logits_left_last=res["left"][:, -1],                                      # the position generate() reads for every row
         logits_right_last=torch.stack([res["right"][i, lens[i] - 1] for i in range(2)]))
    del model.encode_images


def free_running(cfg, build_model, synth):
    # ------------------------------------------------------------------ free-running fp16 sequence with wide margins
    I = -200
    ids1 = [3, I, 17, 18, I, 19, 20, 21, 5, 9]
    n_new, want_margin = 16, 0.03
    best = None
    for seed in range(40, 400):
        mdl, _ = build_model(cfg, seed, torch.float16)
        px16 = T(synth.pixels(2, cfg.vision["image_size"], seed=7)).half()
        o = mdl(input_ids=torch.tensor([ids1]), images=px16, use_cache=True)
        cache, last = o.past_key_values, o.logits[0, -1].float()
        toks, margins = [], []
        for s in range(n_new):
            top2 = torch.topk(last, 2)
            toks.append(int(top2.indices[0])); margins.append(float(top2.values[0] - top2.values[1]))
            o = mdl(input_ids=torch.tensor([[toks[-1]]]), past_key_values=cache, use_cache=True)
            cache, last = o.past_key_values, o.logits[0, -1].float()
        score = min(margins) if len(set(toks)) >= 3 else 0.0        # a sequence stuck on one token says little about the cache
        if best is None or score > best[0]:
            best = (score, seed, toks, margins)
        if score >= want_margin:
            break
    mm, seed, toks, margins = best
    print(f"  free-running fp16 sequence: seed {seed}, min margin {mm:.4f}")
    save("e2e_free_f16", ids=torch.tensor([ids1]), n_tiles=2, pixel_seed=7, seed=seed, tokens=toks, margins=margins)


if __name__ == "__main__":
    main()
