"""Oracle: image-token splice (torch CPU).  Test infrastructure only.

Restates OmChatMetaForCausalLM.prepare_inputs_labels_for_multimodal (omchat/model/omchat_arch.py:55-209;
HF twin omchat/hf/modeling_omchat.py:769-923) for the inference case (labels=None)."""
import torch

IMAGE_TOKEN_INDEX = -200   # omchat/constants.py:9


def splice_inputs(input_ids, attention_mask, image_features, embed_tokens, padding_side="right", max_length=None):
    """Prefill branch (omchat_arch.py:103-209).

    input_ids int64 [b, T] with -200 sentinels; attention_mask [b, T] or None; image_features: sequence of
    [n_tok, H] tensors, one per tile for the WHOLE batch (consumed in row-major order via a running index,
    :119,:149-150; a row without sentinels still consumes one entry as a zero-length slice, :122-129);
    embed_tokens [vocab, H].  Returns (inputs_embeds [b, S, H], attention_mask_out (caller dtype) or None, lengths).
    """
    b, T = input_ids.shape
    _mask = attention_mask
    mask = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
    rows = [ids[m] for ids, m in zip(input_ids, mask)]                     # :115 drop padded ids first
    out_rows = []
    cur = 0
    for ids in rows:
        n_img = int((ids == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:                                                     # :122-129
            out_rows.append(torch.cat([embed_tokens[ids], image_features[cur][0:0].to(embed_tokens.dtype)], dim=0))
            cur += 1
            continue
        pos = [-1] + torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + [ids.shape[0]]
        parts = []
        for i in range(len(pos) - 1):                                      # :133-158
            parts.append(embed_tokens[ids[pos[i] + 1:pos[i + 1]]])
            if i < n_img:
                parts.append(image_features[cur].to(embed_tokens.dtype))
                cur += 1
        out_rows.append(torch.cat(parts, dim=0))
    if max_length is not None:                                             # :161-164
        out_rows = [r[:max_length] for r in out_rows]
    S = max(r.shape[0] for r in out_rows)
    H = embed_tokens.shape[1]
    embeds = torch.zeros(b, S, H, dtype=embed_tokens.dtype)
    new_mask = torch.zeros(b, S, dtype=torch.bool)
    lengths = []
    for i, r in enumerate(out_rows):                                       # :172-195
        n = r.shape[0]
        lengths.append(n)
        if n == 0:
            continue
        if padding_side == "left":
            embeds[i, S - n:] = r
            new_mask[i, S - n:] = True
        else:
            embeds[i, :n] = r
            new_mask[i, :n] = True
    mask_out = None if _mask is None else new_mask.to(_mask.dtype)         # :201-204
    return embeds, mask_out, lengths


def decode_step_inputs(attention_mask, past_len):
    """Decode short-circuit (omchat_arch.py:61-70): extend the mask with ones to past_len+1 and
    position_ids = sum(mask) - 1."""
    target = past_len + 1
    ext = torch.ones((attention_mask.shape[0], target - attention_mask.shape[1]), dtype=attention_mask.dtype)
    m = torch.cat((attention_mask, ext), dim=1)
    return m, torch.sum(m, dim=1).unsqueeze(-1) - 1
