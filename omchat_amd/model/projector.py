"""Counterpart of omchat/model/multimodal_projector/builder.py:39-66."""
import re


class _MLPProjector:
    def __init__(self, engine):
        self.engine = engine

    def forward(self, x):
        if self.engine is None:
            raise RuntimeError("projector has no HIP engine attached (no CPU fallback)")
        return self.engine.projector_forward(x).to(x.dtype)

    __call__ = forward


class IdentityMap:
    def forward(self, x, *args, **kwargs):
        return x

    __call__ = forward

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


def build_vision_projector(config, delay_load=False, engine=None, **kwargs):
    """Same dispatch and the same ValueError as the reference; `mlp2x_gelu` (the released checkpoints' projector,
    convert_omchat_to_hf.py:33-34) runs on the HIP engine."""
    projector_type = getattr(config, "mm_projector_type", "linear")
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        if int(m.group(1)) != 2:
            raise NotImplementedError(f"{projector_type}: only mlp2x_gelu is built for the HIP path")
        return _MLPProjector(engine)
    if projector_type == "identity":
        return IdentityMap()
    if projector_type in ("linear", "cabstract"):
        raise NotImplementedError(f"{projector_type} projector is outside the hot path (SURVEY.md §2 rows 10, 23)")
    raise ValueError(f"Unknown projector type: {projector_type}")
