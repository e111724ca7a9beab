"""Counterpart of omchat/model/language_model/omchat_qwen2.py + omchat/model/omchat_arch.py on the HIP engine."""
import types
import torch

from ..constants import IMAGE_TOKEN_INDEX
from ..config import OmChatConfig
from .vision_tower import build_vision_tower
from .projector import build_vision_projector


class OmChatQwen2Config(OmChatConfig):
    model_type = "omchat_qwen2"


class KVHandle:
    """Opaque handle to the context-owned KV cache.  Quacks like the caches the reference probes:
    `get_seq_length()` (HF Cache) and the legacy `past_key_values[-1][-1].shape[-2]` (omchat_arch.py:63)."""

    def __init__(self, engine, batch):
        self.engine, self.batch = engine, batch

    def get_seq_length(self, layer_idx=0):
        return max(self.engine.kv_lengths(self.batch))

    def __bool__(self):
        return True

    def __getitem__(self, i):
        L = self.get_seq_length()
        kv = self.engine.local["t_kv_heads"]
        probe = types.SimpleNamespace(shape=(self.batch, kv, L, 128))
        return (probe, probe)

    def __len__(self):
        return self.engine.cfg.text["num_hidden_layers"]


class CausalLMOutputWithPast(dict):
    def __init__(self, logits, past_key_values):
        super().__init__(logits=logits, past_key_values=past_key_values, loss=None)
        self.logits, self.past_key_values, self.loss = logits, past_key_values, None


class _GenerationConfig(types.SimpleNamespace):
    pass


class OmChatMetaForCausalLM:
    """omchat_arch.py:42-209."""

    def get_vision_tower(self):
        return self.vision_tower

    def encode_images(self, images):
        """omchat_arch.py:50-53 -- tower then projector, fused in one C ABI call."""
        t = self.get_vision_tower()
        if t.select_feature != "patch":
            return self.mm_projector(t(images))
        return self.engine.encode_images(images, t.select_layer).to(images.dtype)

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images):
        """omchat_arch.py:55-209 for inference (labels pass through as None).  Returns the reference's 6-tuple
        (input_ids|None, position_ids, attention_mask, past_key_values, inputs_embeds|None, labels)."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            if past_key_values is not None and vision_tower is not None and images is not None and input_ids.shape[1] == 1:
                target = past_key_values[-1][-1].shape[-2] + 1                       # :63
                attention_mask = torch.cat((attention_mask, torch.ones((attention_mask.shape[0], target - attention_mask.shape[1]),
                                                                      dtype=attention_mask.dtype, device=attention_mask.device)), dim=1)
                position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if type(images) is list or images.ndim == 5:
            if type(images) is list:
                if any(im.ndim != 3 for im in images):
                    raise NotImplementedError("video inputs: the reference calls an undefined encode_videos (omchat_arch.py:87)")
                images = torch.stack(images)
            else:
                raise NotImplementedError("video inputs: the reference calls an undefined encode_videos (omchat_arch.py:87)")
        if getattr(self.config, "tune_mm_mlp_adapter", False) and getattr(self.config, "mm_use_im_start_end", False):
            raise NotImplementedError                                                  # :100-101
        feats = self.encode_images(images)
        side = getattr(self.config, "tokenizer_padding_side", "right")
        maxlen = getattr(self.config, "tokenizer_model_max_length", None)
        embeds, lengths, valid = self.engine.splice(input_ids, attention_mask, feats, side, maxlen)
        self._last_lengths = lengths
        new_mask = None if attention_mask is None else valid.to(dtype=attention_mask.dtype, device=attention_mask.device)   # :201-204
        new_pos = None                                                                 # :206-207 (None unless the caller passed one)
        if position_ids is not None:
            S = embeds.shape[1]
            new_pos = torch.zeros((embeds.shape[0], S), dtype=position_ids.dtype, device=position_ids.device)
            for i, n in enumerate(lengths):
                if side == "left":
                    new_pos[i, S - n:] = torch.arange(n, dtype=position_ids.dtype)
                else:
                    new_pos[i, :n] = torch.arange(n, dtype=position_ids.dtype)
        return None, new_pos, new_mask, past_key_values, embeds, labels


class OmChatQwen2ForCausalLM(OmChatMetaForCausalLM):
    """OmChatQwen2ForCausalLM (omchat_qwen2.py:29-111): forward(images=...) + HF-style greedy generate()."""

    def __init__(self, config, engine):
        self.config = config
        self.engine = engine
        self.vocab_size = config.text["vocab_size"]
        args = types.SimpleNamespace(**config.mm)
        self.vision_tower = build_vision_tower(args, engine=engine) if engine.c.v_layers > 0 else None
        self.mm_projector = build_vision_projector(args, engine=engine) if engine.c.v_layers > 0 else None
        self.generation_config = _GenerationConfig(pad_token_id=None, eos_token_id=None, max_new_tokens=None)
        self._last_lengths = None
        self.device = engine.device
        self.dtype = engine.torch_dtype

    def get_model(self):
        return self

    def eval(self):
        return self

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None):
        if output_attentions:
            raise NotImplementedError("attention maps are never materialised by the flash kernels")
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values, labels, images)
        if inputs_embeds is None and past_key_values is None:
            # text-only prefill: embed_tokens through the same gather kernel
            inputs_embeds, lengths, _ = self.engine.splice(input_ids, attention_mask, None)
            self._last_lengths = lengths
        if inputs_embeds is not None:
            b, S, _ = inputs_embeds.shape
            lengths = self._last_lengths if self._last_lengths is not None and len(self._last_lengths) == b else [S] * b
            side = "right"
            if attention_mask is not None and attention_mask.shape[1] == S:
                lengths = [int(x) for x in attention_mask.ne(0).sum(dim=1)]
                right = all(int(attention_mask[i, :n].ne(0).sum()) == n for i, n in enumerate(lengths))
                left = all(int(attention_mask[i, S - n:].ne(0).sum()) == n for i, n in enumerate(lengths))
                if not right and not left:
                    raise NotImplementedError("attention masks with holes: pad on one side")
                # a left-padded batch (tokenizer_padding_side='left', omchat_arch.py:176-184) is prefilled exactly as the reference
                # does it (RoPE on arange(S), padded keys masked, logits of position S - 1); its decode steps go through the masked
                # step below (omchat_decode_step_masked), which positions and masks the rows as omchat_arch.py:61-70 does
                side = "right" if right else "left"
            self._last_lengths = None
            # a batch whose rows differ in (spliced) length, or any left-padded one, is decoded as the reference does it: common cache
            # slot, position_ids = sum(mask) - 1 and the padded token-level mask as the key mask (omchat_arch.py:61-70).  That step
            # exists on one GPU only (include/omchat_hip.h: omchat_decode_step_masked): under tensor parallelism a RIGHT-padded ragged
            # batch keeps the per-sequence step (every row at its own length: same tokens as the reference while no row has ended),
            # and a left-padded one is refused HERE, before any work is enqueued
            ragged = b > 1 and (side == "left" and min(lengths) < S or len(set(lengths)) > 1)
            can_mask = self.engine.masked_decode_supported()
            if ragged and side == "left" and not can_mask:
                raise NotImplementedError("left-padded ragged batch: the masked decode step runs at tp_size == 1 only; pad on the right")
            self._padded_batch = ragged and can_mask
            self._prefill_slots = S                                   # the common cache slot a padded batch's decode steps append at
            logits_last, hidden = self.engine.prefill(inputs_embeds, lengths, want_hidden=bool(output_hidden_states), padding_side=side)
            local = logits_last                                       # this rank's vocabulary shard (the whole vocabulary at TP = 1)
            logits_last = self.engine.full_logits(logits_last)        # vocab-parallel lm_head: gather the rank-local shards
            out = CausalLMOutputWithPast(logits_last.unsqueeze(1), KVHandle(self.engine, b))
            out.local_logits = local
            if output_hidden_states:
                out["hidden_states"] = (hidden,); out.hidden_states = (hidden,)
            return out
        if input_ids.shape[1] != 1:
            raise ValueError("decode steps take exactly one token per sequence")
        if getattr(self, "_padded_batch", False):
            if attention_mask is None:
                raise ValueError("decode step of a padded batch: pass the token-level attention_mask (one more 1 per generated token) as HF "
                                 "generate does, so that omchat_arch.py:61-70 can position the rows")
            if position_ids is None:
                # text-only call (no `images`): the reference leaves position_ids to Qwen2Model, which takes cache_position -- the common
                # cache length -- for every row (modeling_qwen2.py:368-373), and the mask is the token-level one as passed
                L = past_key_values.get_seq_length() if past_key_values is not None else max(self.engine.kv_lengths(input_ids.shape[0]))
                position_ids = torch.full((input_ids.shape[0], 1), L, dtype=torch.long)
            nxt, logits = self.engine.decode_step_masked(input_ids[:, 0], position_ids, attention_mask, want_logits=True)
        else:
            nxt, logits = self.engine.decode_step(input_ids[:, 0], want_logits=True)
        out = CausalLMOutputWithPast(self.engine.full_logits(logits).unsqueeze(1), past_key_values)
        out.next_tokens = nxt
        return out

    __call__ = forward

    def _stage_buffer(self, steps, b):
        """pinned staging rows for the generated ids + the event that guards them: allocated once and grown on demand (a hipHostMalloc per
        generate() call was measurable)"""
        cur = getattr(self, "_stage", None)
        if cur is None or cur.shape[0] < steps or cur.shape[1] != b:
            self._stage = torch.empty((max(steps, 64), b), dtype=torch.int32).pin_memory()
            self._stage_event = torch.cuda.Event()
        return self._stage

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """omchat_qwen2.py:92-111."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                             "attention_mask": attention_mask, "images": kwargs.get("images", None)})
        return model_inputs

    @torch.no_grad()
    def generate(self, input_ids=None, images=None, do_sample=False, temperature=0, max_new_tokens=None, streamer=None, use_cache=True,
                 eos_token_id=None, pad_token_id=None, attention_mask=None, stopping_criteria=None, **kwargs):
        """Greedy loop as HF GenerationMixin drives it for single_inference.py:53-62: argmax of the last position (first
        index wins), stop on EOS (kept in the output) or max_new_tokens; returns prompt + new ids [b, T + new]."""
        if do_sample:
            raise NotImplementedError("sampling is outside the hot path; the reference CLIs call generate(do_sample=False)")
        if max_new_tokens is None:
            max_new_tokens = self.generation_config.max_new_tokens or 20
        eos = eos_token_id if eos_token_id is not None else self.generation_config.eos_token_id
        eos = set(eos) if isinstance(eos, (list, tuple)) else ({eos} if eos is not None else set())
        pad = pad_token_id if pad_token_id is not None else self.generation_config.pad_token_id
        b = input_ids.shape[0]
        if streamer is not None:
            streamer.put(input_ids.cpu())
        out = self.forward(input_ids=input_ids, attention_mask=attention_mask, images=images, use_cache=True)
        # ONE rule for the first token at every TP degree: omchat_greedy on this rank's vocabulary shard -- local first-index-wins argmax,
        # then the (max, index) exchange the decode step uses (model.hip: greedy_pick); no torch re-statement on the gathered logits
        tok = self.engine.argmax(out.local_logits)
        padded = getattr(self, "_padded_batch", False)
        # The KV cache is context-owned with a fixed capacity (the reference's DynamicCache grows without bound): generate as
        # many tokens as fit and stop cleanly, returning what was produced, instead of failing mid-stream with 'KV cache full'.
        # A padded batch appends at the COMMON slot (omchat_arch.py:61-70: the cache length, pads included), so its room is counted from
        # there, not from the longest row's own length
        used = self._prefill_slots if padded else max(self.engine.kv_lengths(b))
        room = self.engine.c.max_seq - used + 1
        if max_new_tokens > room:
            import warnings
            warnings.warn(f"max_new_tokens={max_new_tokens} clamped to {room}: KV cache capacity max_seq={self.engine.c.max_seq}")
            max_new_tokens = max(room, 1)
        new = []
        done = torch.zeros(b, dtype=torch.bool)
        # padded batch (rows of different spliced length, or left padding): decoded as the reference does it (omchat_arch.py:61-70).  HF generate
        # passes the token-level mask grown by one 1 per generated token and `images` on every step; the decode branch pads it with ones to the
        # cache length and takes position_ids = sum(mask) - 1.  Over the cache slots that key mask is the same at every step -- [prompt mask |
        # ones] -- and the positions grow by one, so the branch is evaluated ONCE here (this class's mirror of it) and handed to the engine
        # (masked_decode_begin); the steps then need no host data and run ahead of the host exactly like the unpadded ones.
        if padded and max_new_tokens > 1:                             # (one token = the prefill's: no decode step is enqueued, nothing to begin)
            tok_mask = (attention_mask if attention_mask is not None else torch.ones_like(input_ids)).to("cpu", torch.long)
            m1 = torch.cat([tok_mask, torch.ones(b, 1, dtype=torch.long)], dim=1)
            _, pos1, mask1, _, _, _ = self.prepare_inputs_labels_for_multimodal(torch.zeros(b, 1, dtype=torch.long), None, m1, out.past_key_values,
                                                                                None, images)
            if pos1 is None:          # text-only: Qwen2Model positions every row at the common cache length (modeling_qwen2.py:368-373)
                pos1 = torch.full((b, 1), out.past_key_values.get_seq_length(), dtype=torch.long)
            self.engine.masked_decode_begin(pos1, mask1)
        step_fn = self.engine.decode_step_masked_next if padded else self.engine.decode_step
        # The host looks at every token (EOS, streamer, stopping criteria) as the reference's HF loop does -- but not with the GPU idle:
        # token k is copied to pinned host memory behind an event, step k + 1 is enqueued, and only then the host waits for the event.
        # When token k ends the generation the step that was already enqueued is taken back (omchat_kv_rewind).
        stage = self._stage_buffer(max_new_tokens, b)
        evt = self._stage_event
        for step in range(max_new_tokens):
            last = step == max_new_tokens - 1
            stage[step].copy_(tok.to(torch.int32), non_blocking=True)
            evt.record()
            ahead = None
            if not last:
                # rows that already ended are fed the pad id, as HF's loop does (their cache content then equals the reference's too)
                feed = tok if (pad is None or not bool(done.any())) else torch.where(done.to(tok.device), torch.full_like(tok, pad), tok)
                ahead, _ = step_fn(feed)                              # runs while the host handles token `step`
            evt.synchronize()
            t_cpu = stage[step].to(torch.int64)
            if pad is not None:
                t_cpu = torch.where(done, torch.full_like(t_cpu, pad), t_cpu)
            new.append(t_cpu)
            if streamer is not None:
                streamer.put(t_cpu)
            done = done | torch.tensor([int(x) in eos for x in t_cpu])
            stop = bool(done.all()) or last
            if not stop and stopping_criteria:                        # HF StoppingCriteriaList semantics: any criterion stops the batch
                so_far = torch.cat([input_ids.cpu(), torch.stack(new, dim=1)], dim=1)
                stop = any(bool(c(so_far, None)) for c in stopping_criteria)
            if stop:
                if ahead is not None:
                    self.engine.kv_rewind(b, 1)
                break
            tok = ahead
        if streamer is not None:
            streamer.end()
        return torch.cat([input_ids.cpu(), torch.stack(new, dim=1)], dim=1)
