"""Counterpart of omchat/model/builder.py:22-47: load_pretrained_model(...) -> (tokenizer, model, image_processor, context_len).

Reads an OmChat checkpoint directory (config.json + *.safetensors, omchat-native or HF-format keys,
convert_omchat_to_hf.py:26-35) straight into the HIP engine; nothing is instantiated on the CPU."""
import glob
import json
import os

import numpy as np
import torch

from ..config import OmChatConfig, DEFAULT_PINPOINTS
from ..engine import Engine
from .omchat_qwen2 import OmChatQwen2ForCausalLM

_VISION_DEFAULTS = dict(hidden_size=3200, num_attention_heads=25, intermediate_size=12800, num_hidden_layers=45, patch_size=14,
                        image_size=448, layer_norm_eps=1e-6, qk_normalization=True, qkv_bias=False,
                        norm_type="rms_norm")   # configuration_intern_vit.py:63-83 (300m: intern_vit_300m/configuration_intern_vit.py:60-80)


def config_from_json(path):
    with open(os.path.join(path, "config.json")) as f:
        j = json.load(f)
    tj = j.get("text_config", j)                        # HF-format nests the decoder config (hf/configuration_omchat.py:99-198)
    vision = dict(_VISION_DEFAULTS)
    vision.update({k: v for k, v in j.get("vision_config", {}).items() if k in vision})
    text = dict(hidden_size=tj["hidden_size"], num_hidden_layers=tj["num_hidden_layers"], num_attention_heads=tj["num_attention_heads"],
                num_key_value_heads=tj.get("num_key_value_heads", tj["num_attention_heads"]), intermediate_size=tj["intermediate_size"],
                vocab_size=tj["vocab_size"], rms_norm_eps=tj.get("rms_norm_eps", 1e-6),
                rope_theta=tj.get("rope_theta", (tj.get("rope_parameters") or {}).get("rope_theta", 1e6)),
                head_dim=tj.get("head_dim") or tj["hidden_size"] // tj["num_attention_heads"])
    mm = {k: j[k] for k in ("mm_vision_tower", "mm_projector_type", "mm_hidden_size", "mm_vision_select_layer", "mm_vision_select_feature",
                            "image_grid_pinpoints", "tokenizer_padding_side", "tokenizer_model_max_length") if k in j}
    mm.setdefault("image_grid_pinpoints", DEFAULT_PINPOINTS)
    cfg = OmChatConfig(vision, text, mm)
    cfg.max_sequence_length = j.get("max_sequence_length", tj.get("max_sequence_length", 2048))
    cfg.max_position_embeddings = tj.get("max_position_embeddings", j.get("max_position_embeddings"))
    cfg.eos_token_id = j.get("eos_token_id", tj.get("eos_token_id"))
    return cfg


def iter_safetensors(path):
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    for fn in files:
        with safe_open(fn, framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def default_capacity(cfg):
    """(max_seq, max_tiles) when the caller does not size the context.  The reference's DynamicCache has no limit; here the KV
    cache and the prefill workspaces are allocated once, so the default must hold the largest anyres picture the pinpoints allow
    (DEFAULT_PINPOINTS: 3 x 3 tiles + thumbnail = 10 x 1024 image tokens) plus the prompt and single_inference.py's
    max_new_tokens = 1024: the decoder's max_position_embeddings when the checkpoint states it (Qwen2-7B: 32768 -> 1.9 GB of KV per
    sequence), never less than tiles x tokens + 4096."""
    tile = cfg.vision["image_size"]
    pins = cfg.mm.get("image_grid_pinpoints") or DEFAULT_PINPOINTS
    tiles = max((int(w) // tile) * (int(h) // tile) for w, h in pins) + 1
    need = tiles * cfg.num_image_tokens + 4096
    mpe = getattr(cfg, "max_position_embeddings", None)
    max_seq = max(need, min(int(mpe), 32768) if mpe else 16384, int(getattr(cfg, "max_sequence_length", 2048) or 2048))
    return max_seq, max(16, tiles)


def load_omchat_model(model_path, torch_dtype=torch.float16, max_seq=None, max_batch=1, max_tiles=None, tp_rank=0, tp_size=1, comm=None):
    from ..tp import shard_tensor
    from ..weights import prepare_state_dict
    cfg = config_from_json(model_path)
    d_seq, d_tiles = default_capacity(cfg)
    max_seq = max_seq or d_seq
    max_tiles = max_tiles or d_tiles
    eng = Engine(cfg, dtype=torch_dtype, max_seq=max_seq, max_batch=max_batch, max_tiles=max_tiles, max_prefill_rows=max_seq * max_batch,
                 tp_rank=tp_rank, tp_size=tp_size, comm=comm)
    for k, v in iter_safetensors(model_path):           # streamed tensor by tensor: host memory stays small
        for kk, vv in prepare_state_dict({k: v}, cfg).items():
            eng.load_tensor(kk, shard_tensor(kk, vv, cfg, tp_rank, tp_size))
    n = eng.lib.omchat_weights_missing(eng.h)
    if n:
        raise KeyError(eng.lib.omchat_last_error().decode())
    model = OmChatQwen2ForCausalLM(cfg, eng)
    model.generation_config.eos_token_id = getattr(cfg, "eos_token_id", None)
    return model


def load_image_processor(model, device="cuda"):
    """builder.py:42-47."""
    vt = model.get_vision_tower() if hasattr(model, "get_vision_tower") else None
    if vt is not None and not vt.is_loaded:
        vt.load_model(is_train=False)
    return vt.image_processor if vt is not None else None


def load_pretrained_model(model_path, model_name=None, device_map="auto", device="cuda", **kwargs):
    """builder.py:22-35.  `device_map` is accepted for signature compatibility; placement is one context per GPU
    (tensor parallel via tp_rank/tp_size/comm kwargs), not accelerate's layer-wise map."""
    from transformers import AutoTokenizer
    if device != "cuda":
        raise ValueError("the HIP path has no CPU device")
    dtype = kwargs.pop("torch_dtype", torch.float16)      # the reference forces fp16 (builder.py:28)
    try:
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False)
    except Exception:
        tokenizer = AutoTokenizer.from_pretrained(model_path)
    model = load_omchat_model(model_path, torch_dtype=dtype, **{k: v for k, v in kwargs.items()
                                                               if k in ("max_seq", "max_batch", "max_tiles", "tp_rank", "tp_size", "comm")})
    image_processor = load_image_processor(model, device)
    context_len = getattr(model.config, "max_sequence_length", 2048)
    return tokenizer, model, image_processor, context_len


def save_synthetic_checkpoint(path, cfg, seed=0, layout="native", dtype=torch.float16, with_tokenizer=True):
    """Write config.json + model.safetensors (+ a toy tokenizer) with the deterministic synthetic weights, in either key
    layout -- test/demo data for the loader (no real checkpoint is reachable offline)."""
    from safetensors.torch import save_file
    from .. import synth
    from ..weights import to_hf_key
    os.makedirs(path, exist_ok=True)
    sd = synth.state_dict(cfg, seed)
    tensors = {}
    for k, v in sd.items():
        kk = to_hf_key(k) if layout == "hf" else k
        tensors[kk] = torch.from_numpy(v).to(dtype).contiguous()
    save_file(tensors, os.path.join(path, "model.safetensors"))
    t = cfg.text
    tj = dict(hidden_size=t["hidden_size"], num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
              num_key_value_heads=t["num_key_value_heads"], intermediate_size=t["intermediate_size"], vocab_size=t["vocab_size"],
              rms_norm_eps=t["rms_norm_eps"], rope_theta=t["rope_theta"], head_dim=t["head_dim"], max_sequence_length=2048)
    j = dict(model_type="omchat_qwen2", vision_config=cfg.vision, **cfg.mm)
    if layout == "hf":
        j["model_type"] = "omchat"; j["text_config"] = tj
    else:
        j.update(tj)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(j, f)
    if with_tokenizer:
        from tokenizers import Tokenizer, models, pre_tokenizers
        from transformers import PreTrainedTokenizerFast
        vocab = {f"w{i}": i for i in range(min(t["vocab_size"], 300))}
        vocab.update({"\n": len(vocab)})
        tok = Tokenizer(models.WordLevel(vocab, unk_token="w0"))
        tok.pre_tokenizer = pre_tokenizers.Split(" ", "removed")
        PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="w0").save_pretrained(path)
    return path
