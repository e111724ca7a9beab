"""GPU: the reference-shaped Python API (load_pretrained_model -> model.generate(input_ids, images=...), the HF facade,
prepare_inputs_labels_for_multimodal's 6-tuple) on synthetic checkpoints in both key layouts, checked with the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import rel, sync, TOL_DEEP
from omchat_amd import synth
from omchat_amd.config import tiny
from omchat_amd.model import load_pretrained_model, save_synthetic_checkpoint, OmChatForConditionalGeneration, KVHandle
import oracle

T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
I = -200


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory, gpu_lib):
    cfg = tiny()
    root = tmp_path_factory.mktemp("ckpt")
    return cfg, save_synthetic_checkpoint(str(root / "native"), cfg, 9, "native"), save_synthetic_checkpoint(str(root / "hf"), cfg, 9, "hf")


def _oracle_generate(cfg, ids, px, n):
    sd = {k: T32(v) for k, v in synth.state_dict(cfg, 9).items()}
    return oracle.greedy_generate(ids, px, sd, cfg.vision, cfg.text, n)


def test_load_pretrained_model_and_generate(ckpt):
    cfg, native, _ = ckpt
    tokenizer, model, image_processor, context_len = load_pretrained_model(native, "native", max_seq=256, max_tiles=2)
    assert context_len == 2048 and image_processor.crop_size["height"] == 448
    assert model.config.image_grid_pinpoints == cfg.image_grid_pinpoints            # single_inference.py:46
    assert model.get_vision_tower().is_loaded and model.get_vision_tower().num_patches == 16
    model.generation_config.pad_token_id = tokenizer.pad_token_id                   # single_inference.py:50
    px = T32(synth.pixels(2, 56, 5))
    ids = torch.tensor([[3, I, 17, 18, I, 19, 20, 21]])

    class Streamer:
        def __init__(self): self.got, self.ended = [], False
        def put(self, x): self.got.append(x)
        def end(self): self.ended = True
    st = Streamer()
    out = model.generate(ids, images=px.half().cuda(), do_sample=False, temperature=0, max_new_tokens=8, streamer=st, use_cache=True,
                         eos_token_id=151645)
    assert out.shape == (1, ids.shape[1] + 8) and torch.equal(out[:, :ids.shape[1]], ids)
    assert st.ended and len(st.got) == 9 and torch.equal(st.got[0], ids)
    ref, margins = _oracle_generate(cfg, ids, px, 8)
    for i, (a, b, m) in enumerate(zip(out[0, ids.shape[1]:].tolist(), ref, margins)):
        if m < 0.02:
            break                                    # a near-tie in the fp32 oracle: later tokens may legitimately diverge
        assert a == b, (i, a, b, m)
    with pytest.raises(NotImplementedError):
        model.generate(ids, images=px.cuda(), do_sample=True)


def test_generate_stops_on_eos_and_keeps_it(ckpt):
    cfg, native, _ = ckpt
    _, model, _, _ = load_pretrained_model(native, "native", max_seq=256, max_tiles=2)
    px = T32(synth.pixels(1, 56, 6)).half().cuda()
    ids = torch.tensor([[5, I, 7]])
    free = model.generate(ids, images=px, max_new_tokens=6)
    eos = int(free[0, ids.shape[1] + 2])                     # declare the 3rd generated token to be EOS
    out = model.generate(ids, images=px, max_new_tokens=6, eos_token_id=eos)
    first = free[0, ids.shape[1]:].tolist().index(eos)
    assert out.shape[1] == ids.shape[1] + first + 1 and int(out[0, -1]) == eos      # EOS kept (README.md:77)


def test_generate_with_keywords_stopping_criteria(ckpt):
    """KeywordsStoppingCriteria (mm_utils.py:242-274) through generate(stopping_criteria=[...]): stops right after the keyword ids"""
    import types
    from omchat_amd.mm_utils import KeywordsStoppingCriteria
    cfg, native, _ = ckpt
    _, model, _, _ = load_pretrained_model(native, "native", max_seq=256, max_tiles=2)
    px = T32(synth.pixels(1, 56, 6)).half().cuda()
    ids = torch.tensor([[5, I, 7]])
    free = model.generate(ids, images=px, max_new_tokens=8)
    new = free[0, ids.shape[1]:].tolist()
    kw = new[2:4]                                            # the 3rd+4th generated tokens spell the "keyword"

    class _Tok:
        bos_token_id = None
        def __call__(self, s):
            return types.SimpleNamespace(input_ids=kw)
        def batch_decode(self, x, skip_special_tokens=True):
            return [""]
    crit = KeywordsStoppingCriteria(["STOP"], _Tok(), ids)
    out = model.generate(ids, images=px, max_new_tokens=8, stopping_criteria=[crit])
    first = next(i for i in range(1, len(new)) if new[i - 1:i + 1] == kw)
    assert out[0].tolist() == free[0, :ids.shape[1] + first + 1].tolist()


def test_forward_protocol_prefill_then_decode(ckpt):
    """step 0: full ids + images; step >= 1: last token + cache, images re-passed and ignored (omchat_qwen2.py:92-111)"""
    cfg, native, _ = ckpt
    _, model, _, _ = load_pretrained_model(native, "native", max_seq=256, max_tiles=2)
    px = T32(synth.pixels(2, 56, 5))
    ids = torch.tensor([[3, I, 17, I, 19]])
    mask = torch.ones_like(ids)
    r = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, px.half().cuda())
    assert r[0] is None and r[1] is None and r[3] is None and r[5] is None
    assert r[4].shape == (1, 3 + 2 * 16, 256) and r[2].dtype == mask.dtype and int(r[2].sum()) == 35
    out = model(input_ids=ids, attention_mask=mask, images=px.half().cuda(), use_cache=True)
    sd = {k: T32(v) for k, v in synth.state_dict(cfg, 9).items()}
    ref_logits, cache, _ = oracle.prefill(ids, px, sd, cfg.vision, cfg.text)
    sync()
    assert out.logits.shape == (1, 1, 320)
    assert rel(out.logits[0, 0], ref_logits[0, -1]) < TOL_DEEP["f16"]
    past = out.past_key_values
    assert isinstance(past, KVHandle) and past.get_seq_length() == 35 and past[-1][-1].shape[-2] == 35      # omchat_arch.py:63 probe
    tok = int(torch.argmax(out.logits[0, 0]))
    inp = model.prepare_inputs_for_generation(torch.cat([ids, torch.tensor([[tok]])], 1), past_key_values=past, attention_mask=mask,
                                              images=px.half().cuda(), use_cache=True)
    assert inp["input_ids"].shape == (1, 1)
    out2 = model(**inp)
    ref2 = oracle.decode_step(torch.tensor([[tok]]), sd, cfg.text, cache)
    sync()
    assert rel(out2.logits[0, 0], ref2[0, 0]) < TOL_DEEP["f16"]
    assert past.get_seq_length() == 36


def test_hf_facade_same_tokens_as_native(ckpt):
    cfg, native, hf = ckpt
    _, m_native, _, _ = load_pretrained_model(native, "native", max_seq=256, max_tiles=2)
    m_hf = OmChatForConditionalGeneration.from_pretrained(hf, trust_remote_code=True, torch_dtype=torch.float16, max_seq=256, max_tiles=2).cuda()
    assert m_hf.vision_tower.select_layer == -1 and m_hf.multi_modal_projector is not None and m_hf.language_model is not None
    px = T32(synth.pixels(1, 56, 8)).half().cuda()
    inputs = {"input_ids": torch.tensor([[9, I, 4, 2]]), "images": px}
    a = m_hf.generate(**inputs, max_new_tokens=6, do_sample=False, eos_token_id=None, pad_token_id=0)
    b = m_native.generate(inputs["input_ids"], images=px, max_new_tokens=6)
    assert torch.equal(a, b)


def test_hf_example_flow_with_only_the_import_changed(tmp_path, gpu_lib):
    """hf_example.py:7-17 line by line on a synthetic HF-layout checkpoint: AutoModel / AutoProcessor (registered classes),
    BatchFeature with .to('cuda') and attribute access, generate(**inputs, ...), decode of the new ids."""
    from PIL import Image
    from omchat_amd.model.hf import AutoModel, AutoProcessor
    import transformers
    cfg = tiny(vocab=151680)                                  # the ChatML specials 151644 / 151645 (make_context.py:79-80) must be inside the table
    cfg.mm["image_grid_pinpoints"] = [[56, 112], [112, 56], [112, 112]]
    path = save_synthetic_checkpoint(str(tmp_path / "hf"), cfg, 9, "hf")
    model = AutoModel.from_pretrained(path, trust_remote_code=True, torch_dtype=torch.float16, max_seq=512, max_tiles=8).cuda().eval()
    processor = AutoProcessor.from_pretrained(path, trust_remote_code=True)
    image = Image.fromarray(np.random.default_rng(0).integers(0, 256, (50, 100, 3), dtype=np.uint8))
    prompt = "w3 w4 w5"
    inputs = processor(text=prompt, images=image, return_tensors="pt").to("cuda")
    assert isinstance(inputs, transformers.BatchFeature) and inputs.input_ids.is_cuda and inputs["images"].shape[0] == 3
    with torch.inference_mode():
        output_ids = model.generate(**inputs, max_new_tokens=5, do_sample=False, eos_token_id=model.generation_config.eos_token_id,
                                    pad_token_id=processor.tokenizer.pad_token_id)
    assert output_ids.shape[1] == inputs.input_ids.shape[1] + 5
    assert isinstance(processor.tokenizer.decode(output_ids[0, inputs.input_ids.shape[1]:]), str)
    # transformers' own Auto classes resolve the registered config / model / processor classes as well (no auto_map in this checkpoint)
    m2 = transformers.AutoModel.from_pretrained(path, torch_dtype=torch.float16, max_seq=512, max_tiles=8)
    assert type(m2).__name__ == "OmChatForConditionalGeneration"
    out2 = m2.generate(**inputs, max_new_tokens=5, do_sample=False)
    assert torch.equal(out2, output_ids)


def test_generate_clamps_to_kv_capacity_instead_of_failing(ckpt):
    """a long max_new_tokens must not abort mid-stream with 'KV cache full'; what fits is produced and returned"""
    cfg, native, _ = ckpt
    _, model, _, _ = load_pretrained_model(native, "native", max_seq=48, max_tiles=2)
    px = T32(synth.pixels(1, 56, 5)).half().cuda()
    ids = torch.tensor([[3, I, 17, 18]])                      # 3 + 16 = 19 prompt positions -> room for 30 new tokens
    with pytest.warns(UserWarning, match="clamped"):
        out = model.generate(ids, images=px, max_new_tokens=1024, eos_token_id=None)
    assert out.shape[1] == ids.shape[1] + 30
    from omchat_amd.model.builder import default_capacity
    from omchat_amd.config import omchat13b
    big = omchat13b(); big.max_position_embeddings = 32768
    seq, tiles = default_capacity(big)
    assert seq >= 10 * 1024 + 512 + 1024 and tiles >= 10       # the largest anyres picture + prompt + single_inference's 1024 new tokens


def test_text_only_and_bf16(ckpt):
    cfg, native, _ = ckpt
    _, model, _, _ = load_pretrained_model(native, "native", max_seq=128, max_tiles=1, torch_dtype=torch.bfloat16)
    ids = torch.tensor([[3, 4, 5, 6, 7]])
    out = model(input_ids=ids)
    sd = {k: T32(v) for k, v in synth.state_dict(cfg, 9).items()}
    ref, _, _ = oracle.prefill(ids, None, sd, cfg.vision, cfg.text)
    sync()
    assert rel(out.logits[0, 0], ref[0, -1]) < TOL_DEEP["bf16"]


@pytest.mark.parametrize("D", [128, 64])
def test_flash_attention_seam_class(gpu_lib, D):
    """FlashAttention(softmax_scale, attention_dropout).forward(qkv, key_padding_mask, causal, cu_seqlens, max_s, need_weights)
    (intern_vit_6b/flash_attention.py:25-75): the three input forms, the asserts and the (out, None) return"""
    from omchat_amd.model import FlashAttention
    from test_gpu_ops import _attn_ref
    B, S, H = 2, 70, 3
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, S, 3, H, D, generator=g).half()
    fa = FlashAttention(softmax_scale=None)
    out, none = fa(qkv.cuda())
    assert none is None and out.shape == (B, S, H, D)
    q, k, v = qkv.float().unbind(2)
    ref = _attn_ref(q, k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), D ** -0.5, 0, 0, [S] * B)
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(out, ref) < 2.5e-3
    # key_padding_mask (right padding): valid rows equal attention over the valid prefix, padded rows are zero (pad_input)
    lens = [S, 41]
    mask = torch.arange(S)[None, :] < torch.tensor(lens)[:, None]
    outm, _ = fa(qkv.cuda(), key_padding_mask=mask.cuda())
    refm = _attn_ref(q, k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), D ** -0.5, 0, 0, lens)
    assert rel(outm[1, :41], refm[1, :41]) < 2.5e-3 and float(outm[1, 41:].abs().max()) == 0.0 and rel(outm[0], refm[0]) < 2.5e-3
    # packed varlen form (nnz, 3, H, D) + cu_seqlens
    packed = torch.cat([qkv[0, :S], qkv[1, :41]], dim=0).cuda()
    cu = torch.tensor([0, S, S + 41], dtype=torch.int32).cuda()
    outp, _ = fa(packed, cu_seqlens=cu, max_s=S)
    assert rel(outp[:S], refm[0]) < 2.5e-3 and rel(outp[S:], refm[1, :41]) < 2.5e-3
    # causal + explicit scale
    outc, _ = FlashAttention(softmax_scale=0.2)(qkv.cuda(), causal=True)
    refc = _attn_ref(q, k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), 0.2, 1, 0, [S] * B)
    assert rel(outc, refc) < 2.5e-3
    # the reference's asserts (:39-41) and the unsupported mask shape
    with pytest.raises(AssertionError):
        fa(qkv.cuda(), need_weights=True)
    with pytest.raises(AssertionError):
        fa(qkv.cuda().float())
    with pytest.raises(AssertionError):
        fa(qkv)
    holes = mask.clone(); holes[0, 3] = False
    with pytest.raises(NotImplementedError):
        fa(qkv.cuda(), key_padding_mask=holes.cuda())
