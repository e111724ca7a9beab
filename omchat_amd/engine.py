"""Engine: owns one `omchat_ctx` (one GPU / one tensor-parallel rank) and exposes the hot path on torch CUDA tensors.

torch is plumbing here (device memory for inputs/outputs, the current HIP stream); all arithmetic happens inside
libomchat_hip.so.  There is no CPU fallback: constructing an Engine without a GPU raises."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import OmchatConfig, check, ptr, cur_stream
from .config import OmChatConfig


def _torch():
    import torch
    return torch


class Engine:
    def __init__(self, cfg: OmChatConfig, dtype="bf16", max_seq=4096, max_batch=1, max_tiles=4, max_prefill_rows=None,
                 tp_rank=0, tp_size=1, comm=None, device=None, vision=True, text=True):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.OmchatError("omchat_amd.Engine needs a HIP device: the HIP path has no CPU fallback")
        self.lib = _lib.lib()
        self.cfg = cfg
        self.dtype_code = {"f16": _lib.F16, "fp16": _lib.F16, "bf16": _lib.BF16}[dtype] if isinstance(dtype, str) else _lib.dtype_code(dtype)
        self.torch_dtype = _lib.torch_dtype(self.dtype_code)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.tp_rank, self.tp_size = tp_rank, tp_size
        v, t = cfg.vision, cfg.text
        from .tp import local_dims
        ld = local_dims(cfg, tp_rank, tp_size)
        self.local = ld
        c = OmchatConfig()
        c.v_hidden = v["hidden_size"]; c.v_heads = ld["v_heads"]; c.v_qk_channels = v["hidden_size"]; c.v_mlp = ld["v_mlp"]
        c.v_layers = v["num_hidden_layers"] if vision else 0
        c.v_patch = v["patch_size"]; c.v_image = v["image_size"]; c.v_eps = v["layer_norm_eps"]
        c.t_hidden = t["hidden_size"]; c.t_layers = t["num_hidden_layers"] if text else 0
        c.t_heads = ld["t_heads"]; c.t_kv_heads = ld["t_kv_heads"]; c.t_mlp = ld["t_mlp"]; c.t_vocab = ld["t_vocab"]
        c.t_vocab_total = t["vocab_size"]; c.t_eps = t["rms_norm_eps"]; c.rope_theta = t["rope_theta"]
        c.max_seq = max_seq; c.max_batch = max_batch; c.max_tiles = max_tiles
        c.max_prefill_rows = max_prefill_rows if max_prefill_rows is not None else max_seq * max_batch
        c.dtype = self.dtype_code
        c.v_head_dim = v.get("head_dim", 128)
        c.v_norm_type = 1 if v.get("norm_type", "rms_norm") == "layer_norm" else 0
        c.v_no_qk_norm = 0 if v.get("qk_normalization", True) else 1
        self.c = c
        self.ntok = cfg.num_image_tokens
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.omchat_ctx_create(C.byref(c), tp_rank, tp_size, comm, C.byref(h)))
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.omchat_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_tensor(self, name, t):
        """t: torch tensor (CPU or CUDA; fp32 or the engine dtype) or numpy fp32 array, already rank-local."""
        torch = _torch()
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t))
        if t.dtype not in (torch.float32, self.torch_dtype):
            t = t.to(torch.float32)
        t = t.contiguous()
        shape = (C.c_int64 * max(t.dim(), 1))(*(list(t.shape) or [1]))
        check(self.lib.omchat_load_tensor(self.h, name.encode(), ptr(t), shape, max(t.dim(), 1), _lib.dtype_code(t.dtype)))

    def load_state_dict(self, sd, strict=True):
        """sd: omchat-native keys -> full (unsharded) tensors; sharded here for tp_size > 1."""
        from .tp import shard_tensor
        from .weights import prepare_state_dict
        sd = prepare_state_dict(sd, self.cfg, self.c.v_layers > 0, self.c.t_layers > 0)
        for k, v in sd.items():
            self.load_tensor(k, shard_tensor(k, v, self.cfg, self.tp_rank, self.tp_size))
        if strict:
            n = self.lib.omchat_weights_missing(self.h)
            if n:
                raise KeyError(self.lib.omchat_last_error().decode())

    def fill_synthetic(self, seed=0, local=False):
        """local=True (shard profiling, bench.py --shard-of): every tensor of THIS rank's shapes is filled directly -- right geometry,
        not a sharding of the TP = 1 values.
        Deterministic synthetic weights (omchat_amd/synth.py's generator, evaluated on the device).  Under tensor parallelism
        every rank generates each FULL tensor on its GPU and keeps its shard (tp.shard_tensor semantics: zero-padded heads,
        replicated kv heads), so a TP = N group computes the same function as the TP = 1 context filled with the same seed."""
        if self.tp_size == 1 or local:
            check(self.lib.omchat_fill_synthetic(self.h, seed))
            return
        import math
        torch = _torch()
        from . import synth
        from .tp import shard_tensor
        with torch.cuda.device(self.device):
            for key, shape, std, off in synth.tensor_specs(self.cfg):
                vis = key.startswith(synth.TOWER) or key.startswith("model.mm_projector")
                if (vis and self.c.v_layers == 0) or (not vis and self.c.t_layers == 0):
                    continue
                n = int(np.prod(shape))
                full = torch.empty(n, dtype=self.torch_dtype, device=self.device)
                scale = float(np.float32(std)) * math.sqrt(3.0)      # same arithmetic as omchat_fill_synthetic (model.hip)
                check(self.lib.omchat_op_fill_uniform(self.dtype_code, ptr(full), n, (synth.fnv1a64(key) ^ seed) & 0xFFFFFFFFFFFFFFFF,
                                                      scale, off, cur_stream()))
                torch.cuda.current_stream().synchronize()
                self.load_tensor(key, shard_tensor(key, full.view(*shape), self.cfg, self.tp_rank, self.tp_size))
                del full

    def set_noop_allreduce(self):
        """Measurement only (bench.py --shard-of N): the tensor-parallel sums of this context become no-ops, so one rank's share of the
        work runs alone on one GPU.  The outputs are then NOT the model's outputs."""
        fn = C.cast(self.lib.omchat_allreduce_noop, C.c_void_p)
        check(self.lib.omchat_set_allreduce_hook(self.h, fn, None))

    def set_peer(self, peer, max_bytes=0, all_sizes=False):
        """attach a peer all-reduce group member (tp.init_peer) for the tensor-parallel sums of this context"""
        check(self.lib.omchat_ctx_set_peer(self.h, peer, max_bytes, int(all_sizes)))

    def allreduce(self, t):
        """in-place sum of a contiguous CUDA tensor (engine dtype or fp32, byte count a multiple of 16) over the tensor-parallel group"""
        if self.tp_size > 1:
            check(self.lib.omchat_ctx_allreduce(self.h, ptr(t), t.numel(), _lib.dtype_code(t.dtype), cur_stream()))
        return t

    def encode_images_dp(self, tower, pixels, select_layer=-1):
        """Data-parallel vision tower (optional throughput mode): `tower` is a REPLICATED vision-only Engine (tp_size 1)
        on this rank; the tiles are dealt to the ranks in contiguous shares, every rank encodes its share into a zero-filled feature
        buffer and one all-reduce over this (tensor-parallel) engine's transports gathers them: x + 0 is exact, so every rank ends with
        the bits a single tower would have produced.  Replaces ~90 ViT-sized all-reduces per sample by one."""
        torch = _torch()
        n = pixels.shape[0]
        out = torch.zeros(n, self.ntok, self.cfg.text["hidden_size"], dtype=self.torch_dtype, device=self.device)
        lo, hi = self.tp_rank * n // self.tp_size, (self.tp_rank + 1) * n // self.tp_size
        if hi > lo:
            out[lo:hi] = tower.encode_images(pixels[lo:hi], select_layer)
        return self.allreduce(out)

    def comm_stats(self):
        a, b = C.c_long(0), C.c_long(0)
        check(self.lib.omchat_ctx_comm_stats(self.h, C.byref(a), C.byref(b)))
        rs, ag = C.c_long(0), C.c_long(0)
        check(self.lib.omchat_ctx_sp_stats(self.h, C.byref(rs), C.byref(ag)))
        return dict(peer_allreduces=a.value, rccl_allreduces=b.value, sp_reduce_scatters=rs.value, sp_all_gathers=ag.value)

    def enable_fp8_decode(self, on=True):
        """Weight-only OCP e4m3 replica of the decode-streamed decoder weights (quantised on first call); batch-1 decode only."""
        check(self.lib.omchat_enable_fp8_decode(self.h, int(on)))

    def enable_fp8_kv(self, on=True):
        """fp8 (e4m3 + per-position scale) KV cache for the decode steps that follow the NEXT prefill (BASELINE configs[4])"""
        check(self.lib.omchat_enable_fp8_kv(self.h, int(on)))
        self._fp8_kv = bool(on)

    def masked_decode_supported(self):
        """omchat_decode_step_masked (padded-batch decode as omchat_arch.py:61-70 computes it) runs on one GPU; see include/omchat_hip.h"""
        return self.tp_size == 1

    def enable_fp8_prefill(self, on=True):
        """fp8 x fp8 MFMA for the qkv and gate|up GEMMs of the prefill (activations quantised per token, weights per output row)"""
        check(self.lib.omchat_enable_fp8_prefill(self.h, int(on)))

    def enable_decode_graph(self, on=True):
        """Replay each decode step as one hipGraph launch (TP = 1, b <= 32); same kernels and results as the eager step."""
        check(self.lib.omchat_enable_decode_graph(self.h, int(on)))

    def decode_graph_stats(self):
        st, rp, cp = C.c_long(0), C.c_long(0), C.c_long(0)
        check(self.lib.omchat_decode_graph_stats(self.h, C.byref(st), C.byref(rp), C.byref(cp)))
        return dict(steps=st.value, replays=rp.value, captures=cp.value)

    def prof_enable(self, on=True):
        check(self.lib.omchat_prof_enable(self.h, int(on)))

    def prof_read(self, cat, reset=True):
        ms, n = C.c_double(0), C.c_long(0)
        check(self.lib.omchat_prof_read(self.h, cat, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def device_bytes(self):
        return int(self.lib.omchat_device_bytes(self.h))

    # ------------------------------------------------------------------ vision
    def _px(self, pixels):
        torch = _torch()
        if pixels.dim() != 4 or pixels.shape[1] != 3:
            raise ValueError(f"wrong pixel_values size: {tuple(pixels.shape)}")      # modeling_intern_vit.py:337-338
        s = self.cfg.vision["image_size"]
        if pixels.shape[2] != s or pixels.shape[3] != s:
            raise ValueError(f"expected {s}x{s} tiles, got {tuple(pixels.shape[2:])}")
        return pixels.to(device=self.device, dtype=self.torch_dtype).contiguous()

    def vit_forward(self, pixels, select_layer=-1, select_feature="patch"):
        torch = _torch()
        if select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {select_feature}")           # internVIT_encoder.py:42
        px = self._px(pixels)
        n = px.shape[0]
        keep = select_feature == "cls_patch"
        out = torch.empty(n, self.ntok + (1 if keep else 0), self.cfg.vision["hidden_size"], dtype=self.torch_dtype, device=self.device)
        check(self.lib.omchat_vit_forward(self.h, ptr(px), n, select_layer, int(keep), ptr(out), cur_stream()))
        return out

    def projector_forward(self, x):
        torch = _torch()
        x = x.to(device=self.device, dtype=self.torch_dtype).contiguous()
        rows = x.numel() // x.shape[-1]
        out = torch.empty(*x.shape[:-1], self.cfg.text["hidden_size"], dtype=self.torch_dtype, device=self.device)
        check(self.lib.omchat_projector_forward(self.h, ptr(x), rows, ptr(out), cur_stream()))
        return out

    def encode_images(self, pixels, select_layer=-1):
        torch = _torch()
        px = self._px(pixels)
        n = px.shape[0]
        out = torch.empty(n, self.ntok, self.cfg.text["hidden_size"], dtype=self.torch_dtype, device=self.device)
        check(self.lib.omchat_encode_images(self.h, ptr(px), n, select_layer, ptr(out), cur_stream()))
        return out

    # ------------------------------------------------------------------ splice
    def splice(self, input_ids, attention_mask, feats, padding_side="right", max_length=None):
        """input_ids int64 [b,T] (CPU or CUDA), attention_mask [b,T] or None, feats [n_tiles, ntok, H] CUDA or None.
        Returns (inputs_embeds [b,S,H] CUDA, lengths list, valid-mask bool [b,S] CPU)."""
        torch = _torch()
        ids = input_ids.detach().to("cpu", torch.int64).contiguous()
        b, T = ids.shape
        m = None if attention_mask is None else attention_mask.detach().to("cpu").ne(0).to(torch.uint8).contiguous()
        n_tiles = 0 if feats is None else feats.shape[0]
        ntok = self.ntok if feats is None else feats.shape[1]
        side = 1 if padding_side == "left" else 0
        ml = -1 if max_length is None else int(max_length)
        S = C.c_int(0)
        lens = torch.zeros(b, dtype=torch.int32)
        check(self.lib.omchat_splice_plan(ptr(ids), ptr(m), b, T, ntok, n_tiles, side, ml, None, ptr(lens), C.byref(S), self.c.t_vocab_total))
        idx = torch.empty(b, S.value, dtype=torch.int32)
        check(self.lib.omchat_splice_plan(ptr(ids), ptr(m), b, T, ntok, n_tiles, side, ml, ptr(idx), ptr(lens), C.byref(S), 0))
        idx_d = idx.to(self.device)
        f = None if feats is None else feats.to(device=self.device, dtype=self.torch_dtype).contiguous()
        out = torch.empty(b, S.value, self.cfg.text["hidden_size"], dtype=self.torch_dtype, device=self.device)
        check(self.lib.omchat_splice_gather(self.h, ptr(idx_d), ptr(f), ptr(out), b * S.value, cur_stream()))
        return out, [int(x) for x in lens], idx.ne(_lib.PAD_ROW)

    # ------------------------------------------------------------------ decoder
    def prefill(self, embeds, lengths=None, want_hidden=False, want_logits=True, padding_side="right"):
        """padding_side='left': rows hold their tokens at the END (omchat_arch.py:176-184); logits are those of position S - 1 and
        decode steps are refused afterwards (see include/omchat_hip.h: omchat_prefill_left)."""
        torch = _torch()
        e = embeds.to(device=self.device, dtype=self.torch_dtype).contiguous()
        b, S, _ = e.shape
        lens = torch.tensor(lengths if lengths is not None else [S] * b, dtype=torch.int32)
        logits = torch.empty(b, self.c.t_vocab, dtype=torch.float32, device=self.device) if want_logits else None
        hidden = torch.empty(b, S, self.cfg.text["hidden_size"], dtype=self.torch_dtype, device=self.device) if want_hidden else None
        fn = self.lib.omchat_prefill_left if padding_side == "left" else self.lib.omchat_prefill
        check(fn(self.h, ptr(e), b, S, ptr(lens), ptr(logits), ptr(hidden), cur_stream()))
        return logits, hidden

    def decode_step(self, tokens, want_logits=False):
        torch = _torch()
        tk = tokens.to(device=self.device, dtype=torch.int32).contiguous().view(-1)
        b = tk.shape[0]
        logits = torch.empty(b, self.c.t_vocab, dtype=torch.float32, device=self.device) if want_logits else None
        nxt = torch.empty(b, dtype=torch.int32, device=self.device)
        check(self.lib.omchat_decode_step(self.h, ptr(tk), b, ptr(logits), ptr(nxt), cur_stream()))
        return nxt, logits

    def decode_step_masked(self, tokens, position_ids, attention_mask, want_logits=False):
        """Decode step of a padded batch as the reference computes it (omchat_arch.py:61-70): `position_ids` [b] or [b, 1] and
        `attention_mask` [b, slots + 1] are what the decode branch of prepare_inputs_labels_for_multimodal returns (the token-level
        mask padded with ones, sum(mask) - 1).  The new token of every row goes to the common cache slot; see include/omchat_hip.h."""
        torch = _torch()
        tk = tokens.to(device=self.device, dtype=torch.int32).contiguous().view(-1)
        b = tk.shape[0]
        pos = position_ids.detach().to("cpu", torch.int32).contiguous().view(-1)
        m = attention_mask.detach().to("cpu").ne(0).to(torch.uint8).contiguous()
        if pos.shape[0] != b or m.dim() != 2 or m.shape[0] != b:
            raise ValueError("decode_step_masked: position_ids [b], attention_mask [b, slots + 1]")
        logits = torch.empty(b, self.c.t_vocab, dtype=torch.float32, device=self.device) if want_logits else None
        nxt = torch.empty(b, dtype=torch.int32, device=self.device)
        check(self.lib.omchat_decode_step_masked(self.h, ptr(tk), b, ptr(pos), ptr(m), m.shape[1], ptr(logits), ptr(nxt), cur_stream()))
        return nxt, logits

    def masked_decode_begin(self, position_ids, attention_mask):
        """Once after the prefill of a padded batch that HF-style generate will decode: `attention_mask` [b, cols] is the key mask of the
        FIRST decode step as the reference builds it (token-level mask, any ones padding may be left off: slots behind `cols` count as
        visible), `position_ids` [b] its sum(mask) - 1.  decode_step_masked_next then needs no host data (include/omchat_hip.h)."""
        torch = _torch()
        pos = position_ids.detach().to("cpu", torch.int32).contiguous().view(-1)
        m = attention_mask.detach().to("cpu").ne(0).to(torch.uint8).contiguous()
        if m.dim() != 2 or pos.shape[0] != m.shape[0]:
            raise ValueError("masked_decode_begin: position_ids [b], attention_mask [b, cols]")
        cols = min(m.shape[1], self.c.max_seq)
        check(self.lib.omchat_masked_decode_begin(self.h, m.shape[0], ptr(pos), ptr(m), m.shape[1], cols, cur_stream()))

    def decode_step_masked_next(self, tokens, want_logits=False):
        torch = _torch()
        tk = tokens.to(device=self.device, dtype=torch.int32).contiguous().view(-1)
        b = tk.shape[0]
        logits = torch.empty(b, self.c.t_vocab, dtype=torch.float32, device=self.device) if want_logits else None
        nxt = torch.empty(b, dtype=torch.int32, device=self.device)
        check(self.lib.omchat_decode_step_masked_next(self.h, ptr(tk), b, ptr(logits), ptr(nxt), cur_stream()))
        return nxt, logits

    def fused_status(self):
        """(launches, timeout_bits) of the fused attention + o_proj decode launches; timeout_bits != 0 means a hand-off gave up."""
        n, bits = C.c_long(0), C.c_uint(0)
        check(self.lib.omchat_fused_status(self.h, C.byref(n), C.byref(bits)))
        return n.value, bits.value

    def lm_head(self, hidden):
        torch = _torch()
        h = hidden.to(device=self.device, dtype=self.torch_dtype).contiguous()
        n = h.numel() // h.shape[-1]
        out = torch.empty(*h.shape[:-1], self.c.t_vocab, dtype=torch.float32, device=self.device)
        check(self.lib.omchat_lm_head(self.h, ptr(h), n, ptr(out), cur_stream()))
        return out

    def full_logits(self, logits):
        """[b, V / tp] rank-local vocabulary shard -> [b, V] on every rank (HF-style consumers of `out.logits` must not see a
        shard).  Uses the initialised torch.distributed group (the bootstrap channel of tp.init_comm); identity at TP = 1."""
        if self.tp_size == 1 or logits is None:
            return logits
        torch = _torch()
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise _lib.OmchatError("tensor-parallel logits are vocabulary shards: initialise torch.distributed to gather them, or use Engine.argmax")
        cpu = dist.get_backend() == "gloo"
        mine = logits.float().cpu().contiguous() if cpu else logits.float().contiguous()
        parts = [torch.empty_like(mine) for _ in range(self.tp_size)]
        dist.all_gather(parts, mine)
        return torch.cat(parts, dim=-1).to(self.device)

    def kv_lengths(self, b):
        torch = _torch()
        out = torch.zeros(b, dtype=torch.int32)
        check(self.lib.omchat_kv_lengths(self.h, ptr(out), b))
        return [int(x) for x in out]

    def kv_rewind(self, b, n=1):
        """forget the last n decode steps of sequences 0..b-1 (see include/omchat_hip.h: omchat_kv_rewind)"""
        check(self.lib.omchat_kv_rewind(self.h, b, n, cur_stream()))

    def argmax(self, logits):
        torch = _torch()
        out = torch.empty(logits.shape[0], dtype=torch.int32, device=self.device)
        check(self.lib.omchat_greedy(self.h, ptr(logits), logits.shape[0], ptr(out), cur_stream()))
        return out
