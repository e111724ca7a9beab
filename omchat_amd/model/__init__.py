"""Python mirrors of the reference's module boundaries (same names, argument meaning and error behaviour), backed by
the HIP engine.  See INTEGRATION.md for the mapping to reference files."""
from .vision_tower import InternVITVisionTower, InternVIT300mVisionTower, build_vision_tower
from .flash_attention import FlashAttention
from .projector import build_vision_projector
from .omchat_qwen2 import OmChatQwen2ForCausalLM, OmChatQwen2Config, KVHandle
from .builder import load_pretrained_model, save_synthetic_checkpoint
from .hf import OmChatForConditionalGeneration
