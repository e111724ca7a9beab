"""The samples of the full-depth parity tests (tests/test_gpu_fulldepth.py) and of the fixture generator
(tools/make_fulldepth_fixture.py): one definition, so the committed fixture and the GPU run see the same inputs.

  * configs[1]: 3 tiles + 512 text ids -> S = 3584 (the bench.py sample: sentinels laid out as make_context.py:30 does)
  * ragged batch: two rows of one picture each and 40 / 87 text ids -> spliced lengths 1064 / 1111, right-padded
  * configs[3] tower: the 8 tiles of the OmChat-2.1-8B sample
"""
import numpy as np
import torch

from omchat_amd import synth

N_TILES, N_TEXT = 3, 512
RAGGED_TEXT = (40, 87)
N_TILES_300M = 8
FIXTURE = "fulldepth_configs1.npz"
LOGIT_STRIDE = 16          # the fixture keeps every 16th logit of a position (9504 of 152064) + the full norm + the top 8
ACT_STRIDE = 193           # ... and every 193rd element of the tower output / projected features (per tile norms beside it)


def sample():
    px = torch.from_numpy(synth.pixels(N_TILES, 448, 0))
    text = synth.token_ids(N_TEXT, 151643, 1).tolist()
    ids = torch.tensor([[-200, text[0], -200, text[1], -200] + text[2:]])
    return px, ids


def ragged_sample():
    text = synth.token_ids(sum(RAGGED_TEXT), 151643, 7).tolist()
    a, b = text[:RAGGED_TEXT[0]], text[RAGGED_TEXT[0]:]
    rows = [[-200] + a, b[:7] + [-200] + b[7:]]
    T = max(len(r) for r in rows)
    ids = torch.zeros(2, T, dtype=torch.long); mask = torch.zeros(2, T, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.tensor(r); mask[i, :len(r)] = 1
    return ids, mask


def pixels_300m():
    return torch.from_numpy(synth.pixels(N_TILES_300M, 448, 3))


# ---- what the fixture keeps of a tensor, and how a run is compared with it -------------------------------------------
def act_digest(x, stride=ACT_STRIDE):
    """x fp32 [tiles, tokens, C] -> (strided sample of each tile [tiles, m], squared Frobenius norm per tile [tiles])"""
    f = x.reshape(x.shape[0], -1).double()
    return f[:, ::stride].float().numpy(), f.pow(2).sum(1).numpy()


def act_rel(x, sample, norm2, stride=ACT_STRIDE):
    """relative Frobenius error of x against the digest, estimated on the sample: ||x_s - o_s|| / ||o_s|| over all tiles, and per tile.
    (The sample is every `stride`-th element -- tens of thousands of independent entries per tile, so the estimate is within ~1 % of the
    full figure; the full norm is kept to check that the sample's energy is representative.)"""
    f = x.reshape(x.shape[0], -1).double()[:, ::stride]
    o = torch.from_numpy(np.asarray(sample)).double()
    per_tile = ((f - o).pow(2).sum(1) / o.pow(2).sum(1)).sqrt()
    return float(((f - o).pow(2).sum() / o.pow(2).sum()).sqrt()), [float(v) for v in per_tile]


def logit_digest(l, stride=LOGIT_STRIDE):
    """l fp32 [V] -> dict(sample, norm2, top ids / values)"""
    top = torch.topk(l.double(), 8)
    return dict(sample=l[::stride].float().numpy(), norm2=float(l.double().pow(2).sum()),
                top_ids=top.indices.numpy().astype(np.int64), top_vals=top.values.float().numpy())


def logit_rel(l, sample, stride=LOGIT_STRIDE):
    a = l.double()[::stride]
    o = torch.from_numpy(np.asarray(sample)).double()
    return float(((a - o).pow(2).sum() / o.pow(2).sum()).sqrt())
