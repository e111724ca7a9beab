import ctypes as C
import numpy as np
import torch
from omchat_amd import _lib

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
CODE = {"bf16": _lib.BF16, "f16": _lib.F16}
TOL = {"bf16": 1.2e-2, "f16": 2.5e-3}          # one fused op, output rounded to T
TOL_DEEP = {"bf16": 3e-2, "f16": 6e-3}         # several layers


def dev(a, dt):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to("cuda", DT[dt] if isinstance(dt, str) else dt).contiguous()


def rnd(a, dt):
    """round a float array through the 16-bit type (what the device sees), back to fp32 torch CPU"""
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(DT[dt]).float()


def rel(a, b):
    a = a.detach().float().cpu().double(); b = b.detach().float().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def sync():
    torch.cuda.synchronize()


def randn(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def synth_state_dict(cfg, seed=0, keep=None):
    """omchat_amd.synth.state_dict evaluated on the DEVICE (omchat_op_fill_uniform: the same counter-based generator, bit-identical --
    tests/test_gpu_ops.py::test_fill_uniform_bit_exact_with_host_generator) and copied to the host as fp32 numpy arrays.  The numpy generator
    takes ~80 s for one full-width decoder layer; this takes ~1 s.  keep: predicate on the key."""
    import math
    from omchat_amd import synth
    lib = _lib.lib()
    out = {}
    for key, shape, std, off in synth.tensor_specs(cfg):
        if keep is not None and not keep(key):
            continue
        n = int(np.prod(shape))
        t = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        scale = float(np.float32(std)) * math.sqrt(3.0)
        _lib.check(lib.omchat_op_fill_uniform(_lib.BF16, ptr(t), n, (synth.fnv1a64(key) ^ seed) & 0xFFFFFFFFFFFFFFFF, scale, off, None))
        torch.cuda.synchronize()
        out[key] = t.float().cpu().view(*shape).numpy()
    return out
