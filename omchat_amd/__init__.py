"""omchat_amd -- MI355X-native (gfx950) implementation of the OmChat inference hot path.

vision tower (InternViT-6B) -> mlp2x_gelu projector -> image-token splice -> Qwen2 prefill -> greedy decode,
as hand-written HIP kernels behind a C ABI (`include/omchat_hip.h`, `omchat_amd/lib/libomchat_hip.so`) with thin
Python mirrors of the reference's module boundaries."""
from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX
from .config import OmChatConfig, omchat13b, tiny

__all__ = ["IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "OmChatConfig", "omchat13b", "tiny"]
