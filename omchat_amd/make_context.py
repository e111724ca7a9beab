"""Counterpart of omchat/make_context.py: ChatML prompt assembly with one `<image>` sentinel per tile."""
import torch

from .constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN, IM_START_ID, IM_END_ID
from .mm_utils import tokenizer_image_token, process_anyres_image


def make_context(tokenizer, query, history=None, system="", max_window_size=6144, chat_format="chatml"):
    """make_context.py:66-148: <|im_start|>system\\n{system}<|im_end|>\\n ... <|im_start|>user\\n{query}<|im_end|>\\n<|im_start|>assistant\\n,
    history inserted newest-first while it fits max_window_size; returns (raw_text, token ids with -200 sentinels)."""
    history = history or []
    if chat_format == "raw":
        return query, tokenizer.encode(query)
    if chat_format != "chatml":
        raise NotImplementedError(f"Unknown chat format {chat_format!r}")        # make_context.py:146
    im_start, im_end = "<|im_start|>", "<|im_end|>"
    im_start_tokens, im_end_tokens = [IM_START_ID], [IM_END_ID]
    nl_tokens = tokenizer.encode("\n")

    def _tok(role, content):
        if DEFAULT_IMAGE_TOKEN in content:
            return f"{role}\n{content}", tokenizer.encode(role) + nl_tokens + tokenizer_image_token(content, tokenizer, IMAGE_TOKEN_INDEX)
        return f"{role}\n{content}", tokenizer.encode(role) + nl_tokens + tokenizer.encode(content)

    system_text, system_part = _tok("system", system)
    system_tokens = im_start_tokens + system_part + im_end_tokens
    raw_text, context_tokens = "", []
    for turn_query, turn_response in reversed(history):
        q_text, q_part = _tok("user", turn_query)
        r_text, r_part = _tok("assistant", turn_response)
        nxt = nl_tokens + im_start_tokens + q_part + im_end_tokens + nl_tokens + im_start_tokens + r_part + im_end_tokens
        prev = f"\n{im_start}{q_text}{im_end}\n{im_start}{r_text}{im_end}"
        if len(system_tokens) + len(nxt) + len(context_tokens) < max_window_size:
            context_tokens = nxt + context_tokens
            raw_text = prev + raw_text
        else:
            break
    context_tokens = system_tokens + context_tokens
    raw_text = f"{im_start}{system_text}{im_end}" + raw_text
    context_tokens += nl_tokens + im_start_tokens + _tok("user", query)[1] + im_end_tokens + nl_tokens + im_start_tokens + \
        tokenizer.encode("assistant") + nl_tokens
    raw_text += f"\n{im_start}user\n{query}{im_end}\n{im_start}assistant\n"
    return raw_text, context_tokens


def get_context(text, tokenizer, initial_prompt="You are a helpful assistant.", image=None, image_processor=None, image_grid_pinpoints=None,
                device="cuda"):
    """make_context.py:14-43: tiles (thumbnail first) + "<image>\\npatch:<image>...\\n{question}".  The text-only branch of the
    reference reads an undefined name (:37); here it simply uses `text`."""
    if image is not None:
        patches, _ = process_anyres_image(image, image_processor, image_grid_pinpoints, True, return_best_res=True)
        n = len(patches)
        image_tensor = torch.stack(patches, dim=0).half().to(device)
        query = "<image>\n" + "\n".join(["patch:<image>"] * (n - 1)) + "\n" + text.replace("<image>", "").strip()
    else:
        image_tensor = None
        query = text.replace("<image>", "").strip()
    inp, context_tokens = make_context(tokenizer, query, None, initial_prompt)
    return inp, context_tokens, image_tensor
