"""Counterpart of omchat/model/multimodal_encoder/intern_vit_6b/flash_attention.py::FlashAttention (:10-75), the one native
op seam of the reference (it wraps flash-attn's varlen kernel).  Same constructor, same forward signature, same asserts, same
`(output, None)` return; the attention itself is libomchat_hip.so's flash kernel (head_dim 128 or 64)."""
import ctypes as C
import torch

from .. import _lib
from .._lib import check, ptr, cur_stream


class FlashAttention:
    def __init__(self, softmax_scale=None, attention_dropout=0.0, device=None, dtype=None):
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout
        self.training = False

    def eval(self):
        return self

    def _run(self, qkv, seqlens, causal):
        B, S, three, H, D = qkv.shape
        if three != 3:
            raise ValueError(f"qkv must be [B, S, 3, H, D], got {tuple(qkv.shape)}")
        q = qkv.contiguous()
        out = torch.empty(B, S, H, D, dtype=q.dtype, device=q.device)
        with torch.cuda.device(q.device):
            check(_lib.lib().omchat_mha_fwd_varlen(ptr(q), B, S, H, D, ptr(seqlens), C.c_float(self.softmax_scale or 0.0), int(bool(causal)),
                                                   ptr(out), _lib.dtype_code(q.dtype), cur_stream()))
        return out

    def forward(self, qkv, key_padding_mask=None, causal=False, cu_seqlens=None, max_s=None, need_weights=False):
        """qkv (B, S, 3, H, D), or (nnz, 3, H, D) with cu_seqlens; key_padding_mask bool (B, S).  Inference only (dropout is
        applied by the reference in training mode only, :50,60,70)."""
        assert not need_weights
        assert qkv.dtype in [torch.float16, torch.bfloat16]
        assert qkv.is_cuda
        if cu_seqlens is None:
            if key_padding_mask is None:
                return self._run(qkv, None, causal), None
            # :56-67 unpad -> varlen kernel -> pad_input (zeros at the padded positions).  Right-padded masks map onto per-sequence
            # key lengths; a mask with holes would need the gather/scatter of unpad_input and is not a shape the towers produce
            m = key_padding_mask.to(torch.bool)
            lens = m.sum(dim=1).to(torch.int32)
            S = m.shape[1]
            prefix = torch.arange(S, device=m.device)[None, :] < lens[:, None]
            if not bool((m == prefix).all()):
                raise NotImplementedError("key_padding_mask must be a right-padding mask (valid tokens first)")
            if bool((lens == 0).any()):
                raise ValueError("every sequence needs at least one valid token")
            out = self._run(qkv, lens.to(qkv.device).contiguous(), causal)
            return out * m[:, :, None, None].to(out.dtype), None
        assert max_s is not None
        # packed varlen input (nnz, 3, H, D): one launch per sequence over its row range (host reads cu_seqlens once)
        cu = [int(x) for x in cu_seqlens.tolist()]
        out = torch.empty(qkv.shape[0], qkv.shape[2], qkv.shape[3], dtype=qkv.dtype, device=qkv.device)
        for i in range(len(cu) - 1):
            a, b = cu[i], cu[i + 1]
            if b > a:
                out[a:b] = self._run(qkv[a:b].unsqueeze(0), None, causal)[0]
        return out, None

    __call__ = forward
