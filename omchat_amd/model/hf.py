"""HF-format façade: counterpart of omchat/hf/modeling_omchat.py::OmChatForConditionalGeneration (:677-1353) with
the `vision_tower / multi_modal_projector / language_model` submodule names.  Same math as the native path (the
reference's HF splice :769-923 is line-for-line the native one); only the checkpoint key layout and the fixed
select layer (-1, :750-753) differ."""
import torch

from .builder import load_omchat_model


class OmChatForConditionalGeneration:
    def __init__(self, model):
        self._m = model
        self.config = model.config
        self.generation_config = model.generation_config
        model.vision_tower.select_layer = -1
        self.vision_tower = model.vision_tower
        self.multi_modal_projector = model.mm_projector
        self.language_model = model

    @classmethod
    def from_pretrained(cls, path, trust_remote_code=True, torch_dtype=torch.float16, **kw):
        return cls(load_omchat_model(path, torch_dtype=torch_dtype, **kw))

    def cuda(self):
        return self

    def eval(self):
        return self

    def forward(self, input_ids=None, images=None, **kw):
        return self._m.forward(input_ids=input_ids, images=images, **kw)

    __call__ = forward

    def generate(self, input_ids=None, images=None, **kw):
        """hf_example.py:16: model.generate(**inputs, max_new_tokens, do_sample, eos_token_id, pad_token_id)."""
        return self._m.generate(input_ids=input_ids, images=images, **kw)
