"""HF-format entry: counterpart of omchat/hf/modeling_omchat.py::OmChatForConditionalGeneration (:677-1353) with the
`vision_tower / multi_modal_projector / language_model` submodule names, registered with transformers' Auto classes the way the
reference registers its own (omchat/model/language_model/omchat_qwen2.py:113-114), so that hf_example.py:7-17 runs with only
its import line changed:

    from omchat_amd.model.hf import AutoModel, AutoProcessor          # was: from transformers import AutoModel, AutoProcessor
    model = AutoModel.from_pretrained(path, trust_remote_code=True, torch_dtype=torch.float16).cuda().eval()
    processor = AutoProcessor.from_pretrained(path, trust_remote_code=True)
    inputs = processor(text=prompt, images=image, return_tensors="pt").to("cuda")
    output_ids = model.generate(**inputs, max_new_tokens=1024, do_sample=False, eos_token_id=model.generation_config.eos_token_id, ...)
    processor.tokenizer.decode(output_ids[0, inputs.input_ids.shape[1]:])

Same math as the native path (the reference's HF splice :769-923 is line-for-line the native one); only the checkpoint key layout
and the fixed select layer (-1, :750-753) differ.  `AutoModel` / `AutoProcessor` exported here are transformers' own classes after
registration; `from_pretrained` is wrapped only to drop `trust_remote_code` (a real OmChat checkpoint carries an `auto_map` that
would otherwise pull the reference's PyTorch modelling code from the checkpoint directory instead of this library)."""
import json
import os

import torch
from transformers import AutoConfig as _HFAutoConfig, AutoModel as _HFAutoModel, AutoModelForCausalLM as _HFAutoCausal
from transformers import AutoProcessor as _HFAutoProcessor, PretrainedConfig

from .builder import load_omchat_model


class OmChatHFConfig(PretrainedConfig):
    """hf/configuration_omchat.py:99-198 (model_type 'omchat'): carries config.json verbatim; the engine geometry is re-read from
    the checkpoint directory by builder.config_from_json."""
    model_type = "omchat"

    def __init__(self, vision_config=None, text_config=None, **kw):
        self.vision_config = vision_config or {}
        self.text_config = text_config or {}
        super().__init__(**kw)


class OmChatQwen2HFConfig(PretrainedConfig):
    """omchat-native checkpoints (model_type 'omchat_qwen2', omchat_qwen2.py:18-19)."""
    model_type = "omchat_qwen2"


class OmChatForConditionalGeneration:
    config_class = OmChatHFConfig

    def __init__(self, model):
        self._m = model
        self.config = model.config
        self.generation_config = model.generation_config
        model.vision_tower.select_layer = -1
        self.vision_tower = model.vision_tower
        self.multi_modal_projector = model.mm_projector
        self.language_model = model
        self.device = model.device
        self.dtype = model.dtype

    @classmethod
    def from_pretrained(cls, path, *model_args, config=None, trust_remote_code=True, torch_dtype=torch.float16, dtype=None, **kw):
        keep = {k: v for k, v in kw.items() if k in ("max_seq", "max_batch", "max_tiles", "tp_rank", "tp_size", "comm")}
        return cls(load_omchat_model(path, torch_dtype=dtype or torch_dtype, **keep))

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def forward(self, input_ids=None, images=None, **kw):
        return self._m.forward(input_ids=input_ids, images=images, **kw)

    __call__ = forward

    def generate(self, input_ids=None, images=None, **kw):
        """hf_example.py:16: model.generate(**inputs, max_new_tokens, do_sample, eos_token_id, pad_token_id)."""
        return self._m.generate(input_ids=input_ids, images=images, **kw)


class OmChatQwen2ForCausalLMHF(OmChatForConditionalGeneration):
    """AutoModelForCausalLM entry of the omchat-native layout (omchat_qwen2.py:114): same engine, caller-chosen select layer."""
    config_class = OmChatQwen2HFConfig

    def __init__(self, model):
        sl = model.vision_tower.select_layer
        super().__init__(model)
        model.vision_tower.select_layer = sl


def _register():
    for name, cfg_cls in (("omchat", OmChatHFConfig), ("omchat_qwen2", OmChatQwen2HFConfig)):
        try:
            _HFAutoConfig.register(name, cfg_cls, exist_ok=True)
        except TypeError:          # older transformers: no exist_ok
            try:
                _HFAutoConfig.register(name, cfg_cls)
            except ValueError:
                pass
    for auto, cfg_cls, model_cls in ((_HFAutoModel, OmChatHFConfig, OmChatForConditionalGeneration),
                                     (_HFAutoCausal, OmChatQwen2HFConfig, OmChatQwen2ForCausalLMHF),
                                     (_HFAutoModel, OmChatQwen2HFConfig, OmChatQwen2ForCausalLMHF)):
        try:
            auto.register(cfg_cls, model_cls, exist_ok=True)
        except TypeError:
            try:
                auto.register(cfg_cls, model_cls)
            except ValueError:
                pass
    from ..processing import OmChatProcessor
    for cfg_cls in (OmChatHFConfig, OmChatQwen2HFConfig):
        try:
            _HFAutoProcessor.register(cfg_cls, OmChatProcessor, exist_ok=True)
        except TypeError:
            try:
                _HFAutoProcessor.register(cfg_cls, OmChatProcessor)
            except ValueError:
                pass


_register()


def _model_type(path):
    with open(os.path.join(path, "config.json")) as f:
        return json.load(f).get("model_type", "omchat")


class AutoModel:
    """transformers.AutoModel with the OmChat classes registered; `trust_remote_code` is accepted and ignored (module docstring)."""

    @staticmethod
    def from_pretrained(path, *args, trust_remote_code=None, **kw):
        if os.path.isdir(path) and _model_type(path) in ("omchat", "omchat_qwen2"):
            cls = OmChatForConditionalGeneration if _model_type(path) == "omchat" else OmChatQwen2ForCausalLMHF
            return cls.from_pretrained(path, *args, **kw)
        return _HFAutoModel.from_pretrained(path, *args, trust_remote_code=trust_remote_code, **kw)

    register = _HFAutoModel.register


class AutoProcessor:
    @staticmethod
    def from_pretrained(path, *args, trust_remote_code=None, **kw):
        if os.path.isdir(path) and _model_type(path) in ("omchat", "omchat_qwen2"):
            from ..processing import OmChatProcessor
            return OmChatProcessor.from_pretrained(path, **kw)
        return _HFAutoProcessor.from_pretrained(path, *args, trust_remote_code=trust_remote_code, **kw)

    register = _HFAutoProcessor.register
