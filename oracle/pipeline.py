"""Oracle: whole hot path (encode_images -> splice -> prefill -> greedy decode).  Test infrastructure only.

The state dict uses the reference's omchat-native key layout:
`model.vision_tower.vision_tower.*`, `model.mm_projector.{0,2}.*`, `model.embed_tokens.weight`,
`model.layers.*`, `model.norm.weight`, `lm_head.weight`."""
import torch
from .vit import vision_tower_forward, projector_forward
from .decoder import qwen2_model, lm_head, KVCache
from .splice import splice_inputs

TOWER_PFX = "model.vision_tower.vision_tower."
PROJ_PFX = "model.mm_projector."


def _sub(sd, pfx):
    return {k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}


def encode_images(images, sd, vcfg, select_layer=-1, select_feature="patch"):
    """encode_images (omchat_arch.py:50-53) = tower (internVIT_encoder.py:45-56) then mm_projector."""
    feats = vision_tower_forward(images, _sub(sd, TOWER_PFX), vcfg, select_layer, select_feature)
    return projector_forward(feats, _sub(sd, PROJ_PFX))


def prefill(input_ids, images, sd, vcfg, tcfg, attention_mask=None, select_layer=-1, padding_side="right"):
    """Step 0 of OmChatQwen2ForCausalLM.forward (omchat_qwen2.py:45-89): splice then Qwen2 forward with a fresh
    cache; returns (logits [b, S, vocab], cache, lengths)."""
    emb = sd["model.embed_tokens.weight"]
    if images is not None:
        feats = encode_images(images, sd, vcfg, select_layer)
        feats = [f for f in feats]
    else:
        feats = []
    if images is None:
        embeds = emb[input_ids]
        lengths = [input_ids.shape[1]] * input_ids.shape[0]
        mask = attention_mask
    else:
        embeds, mask, lengths = splice_inputs(input_ids, attention_mask, feats, emb, padding_side,
                                              tcfg.get("tokenizer_model_max_length"))
    cache = KVCache(tcfg["num_hidden_layers"])
    h = qwen2_model(embeds, sd, tcfg, cache, None, mask)
    return lm_head(h, sd), cache, lengths


def decode_step(token_ids, sd, tcfg, cache, attention_mask=None):
    """Step >= 1: last token + cache (omchat_qwen2.py:92-111; omchat_arch.py:61-70)."""
    embeds = sd["model.embed_tokens.weight"][token_ids]            # [b, 1, H]
    h = qwen2_model(embeds, sd, tcfg, cache, None, attention_mask)
    return lm_head(h, sd)


def greedy_generate(input_ids, images, sd, vcfg, tcfg, max_new_tokens, eos_token_id=None, select_layer=-1):
    """HF GenerationMixin greedy loop as driven by single_inference.py:53-62 (b=1): argmax over the last
    position's logits (first index wins ties), stop on EOS (kept in the output) or max_new_tokens.
    Returns (new ids list, list of (top1-top2) fp32 logit margins)."""
    logits, cache, _ = prefill(input_ids, images, sd, vcfg, tcfg)
    out, margins = [], []
    for _ in range(max_new_tokens):
        last = logits[0, -1].float()
        top2 = torch.topk(last, 2)
        nxt = int(torch.argmax(last))
        out.append(nxt)
        margins.append(float(top2.values[0] - top2.values[1]))
        if eos_token_id is not None and nxt == eos_token_id:
            break
        logits = decode_step(torch.tensor([[nxt]]), sd, tcfg, cache)
    return out, margins
