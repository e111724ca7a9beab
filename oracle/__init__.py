"""CPU oracle for the OmChat hot path (vision tower -> projector -> splice -> Qwen2 prefill/decode).

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-CPU restatement of the reference's algorithm
(om-ai-lab/OmChat @ 2024-10-22 plus the `transformers` Qwen2 decoder it inherits from).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and only as the checker.
The product path (`omchat_amd/`) never imports it and fails loudly when the HIP library is missing.

Parity pin: every function here is checked against golden vectors captured from the imported reference
(`tools/make_golden.py`, fixtures in `tests/golden/`).  The decoder arithmetic lives in the third-party
`transformers` package (reference pins ==4.41.2, pyproject.toml:22; this container has 5.15.0): its
formulas are restated from `transformers/models/qwen2/modeling_qwen2.py` and pinned by golden vectors
generated through the reference's own `OmChatQwen2ForCausalLM` wrapper with `attn_implementation="eager"`.
The Qwen2 tokenizer (prompt -> ids) is NOT covered: vocab ships with the checkpoint (absent) -> "parity unpinned"
for that boundary only.
"""
from .vit import (rms_norm, vit_embeddings, vit_attention, vit_mlp, vit_layer, vit_encoder,
                  vision_tower_forward, projector_forward)
from .decoder import (rope_cos_sin, apply_rope, qwen2_attention, qwen2_mlp, qwen2_layer, qwen2_model,
                      lm_head, KVCache)
from .splice import splice_inputs, decode_step_inputs
from .pipeline import encode_images, prefill, decode_step, greedy_generate
