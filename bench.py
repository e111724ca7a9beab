#!/usr/bin/env python3
"""bench.py -- OmChat hot path on MI355X: ViT tiles -> projector -> splice -> prefill -> greedy decode.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gen G] [--workload both|configs1|configs2|configs3|configs4]
    python bench.py --shard-of 8 [...]            # ONE rank of a TP = 8 group on one GPU, exchanges removed (kernel time per rank)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process never touches the GPU (not even to count devices); it
starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each), relays rank 0's JSON line and exits with the
worst exit code.  Under torchrun the ranks are already there.  Ranks form ONE tensor-parallel group (Megatron TP, omchat_amd/tp.py):
the all-reduces run on RCCL over xGMI (large messages) and on the peer one-shot kernel of csrc/comm.hip (decode-sized ones).

Workloads (BASELINE.json `configs`):
  configs1  one sample = one 448x448 picture => 3 anyres tiles (mm_utils.py:28-37,151) + 512 text ids => S = 3584 prefill
            tokens, then G greedy tokens, batch 1.  `value` = generated tokens / s over whole steps (prefill included).
  configs2  32 such samples per step: 96 tiles through the ViT, right-padded batch prefill of 32 x 3584 rows, G batched
            decode steps.  Reported under "configs2" (tokens / s, samples / s, HBM fraction of the decode steps); it is `value`
            only with --workload configs2.
  configs3  OmChat-2.1-8B (InternViT-300M + Qwen2-7B): one sample = 8 pictures of 448x448, each through the dynamic tiling of
            mm_utils.py:276-323 (device front end, bit-exact) -> 8 tiles in ONE batched tower pass, prefill of 8 x 1024 + 512 = 8704
            positions, G greedy tokens.  Side block "configs3" of the default run (own context); `value` with --workload configs3.
  configs4  one 32-frame clip: 32 tiles through the ViT, prefill of 32 x 1024 + 512 = 33 280 positions (>= 16 k visual + text
            tokens), greedy tokens with the fp8 modes on: e4m3 decode weights, e4m3 KV cache, fp8 x fp8 MFMA for the qkv / gate|up
            prefill GEMMs.  A quantised computation: not comparable with `value` of configs1.  Side block "configs4" of the default
            run (own context, 64 tokens); `value` with --workload configs4.
Inputs are resident in HBM before the timed region.  Weights: deterministic synthetic (omchat_amd/synth.py) at the full
geometry, generated on the device; under TP every rank keeps its SHARD of the same values, and the first 32 greedy ids are
checked against a TP = 1 context that rank 0 runs first in the same process ("tokens_match_tp1").

Prints ONE JSON line (rank 0): `roofline` = dominant kernel of the step (decode gate|up weight stream, HBM-bound) from HIP
events on the launch stream inside the timed region; `cpu_baseline` = the oracle on a bounded sample (N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/f16 MFMA
MFMA_PEAK_FP8_TFLOPS = 5000.0    # dense fp8 MFMA (reached by the block-scaled opcodes only, MI355X_MICROARCH.md Matrix cores)
TP1_CHECK_TOKENS = 32
MARGIN_GUARD = {"bf16": 0.05, "f16": 0.02}      # top-1 / top-2 logit gap below which 16-bit rounding may legitimately flip an id


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="both", choices=["both", "configs1", "configs2", "configs3", "configs4"])
    ap.add_argument("--frames", type=int, default=32, help="configs4: video frames (one 448x448 tile = 1024 visual tokens each)")
    ap.add_argument("--images", type=int, default=8, help="configs3: pictures per sample")
    ap.add_argument("--graph", action="store_true", help="replay each decode step as one captured hipGraph instead of ~230 eager launches "
                    "(measured SLOWER on ROCm 7.2 / MI355X: 3.21 vs 2.96 ms per token, so it is off by default)")
    ap.add_argument("--no-fp8", action="store_true", help="skip the (untimed) weight-only fp8 decode measurement")
    ap.add_argument("--no-side", action="store_true", help="default workload: skip the configs3 / configs4 side blocks")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gen", type=int, default=256, help="greedy decode tokens per step")
    ap.add_argument("--batch2", type=int, default=32, help="samples per step of the configs2 workload")
    ap.add_argument("--steps2", type=int, default=0, help="timed steps of the configs2 side measurement (0 = min(steps, 3))")
    ap.add_argument("--text-tokens", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "peer"],
                    help="tensor-parallel all-reduce: auto = RCCL for large + peer one-shot (csrc/comm.hip) for <= 256 KiB messages")
    ap.add_argument("--vit", default="both", choices=["tp", "dp", "both"],
                    help="N > 1: vision tower tensor-parallel (north star; the headline), data-parallel over the tiles with a replicated tower and "
                         "one gather (SURVEY 8e optional throughput mode), or tp as the headline with dp measured beside it")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N",
                    help="N = 1 GPU only: run rank 0's share of a TP = N group (its shard shapes and launch sequence) with the exchanges removed "
                         "(omchat_allreduce_noop): per-rank kernel time without an N-GPU node.  The outputs are not the model's outputs; "
                         "the line carries shard_of and is not a throughput claim")
    ap.add_argument("--no-tp1-check", action="store_true")
    ap.add_argument("--tp1-check-dtype", default="", choices=["", "bf16", "f16"],
                    help="N > 1: dtype of the TP = N vs TP = 1 logit / id cross-check contexts (default f16: every position clears the margin "
                         "guard there; the timed run keeps --dtype)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny geometry (NOT the benchmark config)")
    ap.add_argument("--tuning", default="", help="debug: comma-separated key=value pairs for omchat_op_set_tuning (include/omchat_hip.h)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (this process never initialises HIP)
# ----------------------------------------------------------------------------------------------------------------------
def count_gpus_without_hip():
    """GPUs of this node from the KFD topology (nodes with SIMDs), honouring HIP / ROCR_VISIBLE_DEVICES: the spawning parent must never
    reach the HIP runtime (its children are started from it), and it does not need to."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(a):
    import socket
    import tempfile
    ndev = count_gpus_without_hip()
    oversub = os.environ.get("OMCHAT_BENCH_OVERSUBSCRIBE") == "1"
    if ndev < a.gpus and not oversub:
        print(json.dumps({"error": f"--gpus {a.gpus} but only {ndev} GPU(s) visible", "n_gpus": a.gpus}))
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    logdir = tempfile.mkdtemp(prefix="omchat_bench_")
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if r == 0 else open(os.path.join(logdir, f"rank{r}.out"), "w")
        err = open(os.path.join(logdir, f"rank{r}.err"), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err), err.name))
    out0, _ = procs[0][0].communicate()
    rcs = [p.wait() for p, _ in procs]
    text = out0.decode(errors="replace") if out0 else ""
    line = next((l for l in reversed(text.splitlines()) if l.startswith("{")), None)
    if line:
        print(line)
    if any(rcs) or not line:
        for (p, errf), rc in zip(procs, rcs):
            try:
                tail = open(errf).read()[-3000:]
            except OSError:
                tail = ""
            sys.stderr.write(f"---- rank exit code {rc}: {errf}\n{tail}\n")
        if not line:
            print(json.dumps({"error": "no rank-0 result", "n_gpus": a.gpus, "exit_codes": rcs}))
    return max(abs(rc) for rc in rcs) if any(rcs) else (0 if line else 1)


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo restricted to this process's affinity mask; os.cpu_count() when that cannot be read"""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, val = [x.strip() for x in line.split(":", 1)]
                cur[k] = val
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
        return max(1, len(seen))
    except (OSError, ValueError, AttributeError):
        return os.cpu_count() or 1


def oracle_full_depth_record():
    """wall time per phase of the LIVE full-depth oracle pass on an MI355X box's host (tests/test_gpu_fulldepth.py with OMCHAT_LIVE_ORACLE=1,
    committed under profiles/): the un-extrapolated CPU figure beside the bounded sample this run times"""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_oracle_cpu_phases.json")), key=_profile_key)
    if not files:
        return None
    d = json.load(open(files[-1]))
    d["file"] = "profiles/" + os.path.basename(files[-1])
    return d


def cpu_baseline(cfg, S, gen, n_tiles):
    """The oracle (kind 'port') on this host's cores, bounded sample: 1 ViT layer on 1 tile (1025 tokens), 1 decoder
    layer prefill at S, 4 decode steps of 1 decoder layer at L = S, lm_head once; extrapolated to the whole step."""
    import torch
    import oracle
    from oracle.decoder import qwen2_layer, rope_cos_sin
    torch.manual_seed(0)
    # threads = ALL PHYSICAL cores of this host, stated (rounds 1-5 took whatever count a matvec probe
    # preferred -- 32 on one box, 128 on another: a 2.3 x spread of the baseline on the same code).  SMT siblings are left idle.
    ncpu = os.cpu_count() or 1
    cores = physical_cores()
    torch.set_num_threads(cores)
    v, t = cfg.vision, cfg.text
    C, I = v["hidden_size"], v["intermediate_size"]
    g = lambda *s: torch.randn(*s) * 0.02
    w = {"encoder.layers.0.ls1": g(C) + 0.1, "encoder.layers.0.ls2": g(C) + 0.1, "encoder.layers.0.norm1.weight": torch.ones(C),
         "encoder.layers.0.norm2.weight": torch.ones(C), "encoder.layers.0.attn.qkv.weight": g(3 * C, C),
         "encoder.layers.0.attn.q_norm.weight": torch.ones(C), "encoder.layers.0.attn.k_norm.weight": torch.ones(C),
         "encoder.layers.0.attn.proj.weight": g(C, C), "encoder.layers.0.attn.proj.bias": g(C),
         "encoder.layers.0.mlp.fc1.weight": g(I, C), "encoder.layers.0.mlp.fc1.bias": g(I),
         "encoder.layers.0.mlp.fc2.weight": g(C, I), "encoder.layers.0.mlp.fc2.bias": g(C)}
    ntok = cfg.num_image_tokens + 1
    x = torch.randn(1, ntok, C)
    with torch.no_grad():
        oracle.vit_layer(x[:, :65], w, 0, v["num_attention_heads"])           # touch pages
        t0 = time.perf_counter(); oracle.vit_layer(x, w, 0, v["num_attention_heads"]); t_vit = time.perf_counter() - t0
        H, It = t["hidden_size"], t["intermediate_size"]
        nh, nkv, d = t["num_attention_heads"], t["num_key_value_heads"], t["head_dim"]
        P = "model.layers.0."
        wd = {P + "self_attn.q_proj.weight": g(nh * d, H), P + "self_attn.q_proj.bias": g(nh * d),
              P + "self_attn.k_proj.weight": g(nkv * d, H), P + "self_attn.k_proj.bias": g(nkv * d),
              P + "self_attn.v_proj.weight": g(nkv * d, H), P + "self_attn.v_proj.bias": g(nkv * d),
              P + "self_attn.o_proj.weight": g(H, nh * d), P + "mlp.gate_proj.weight": g(It, H), P + "mlp.up_proj.weight": g(It, H),
              P + "mlp.down_proj.weight": g(H, It), P + "input_layernorm.weight": torch.ones(H),
              P + "post_attention_layernorm.weight": torch.ones(H)}
        xe = torch.randn(1, S, H) * 0.5
        cache = oracle.KVCache(1)
        cos, sin = rope_cos_sin(torch.arange(S)[None], d, t["rope_theta"], torch.float32)
        t0 = time.perf_counter(); qwen2_layer(xe, wd, 0, t, cos, sin, cache); t_pre = time.perf_counter() - t0
        nstep = 4
        t0 = time.perf_counter()
        for i in range(nstep):
            c1, s1 = rope_cos_sin(torch.tensor([[S + i]]), d, t["rope_theta"], torch.float32)
            qwen2_layer(xe[:, :1], wd, 0, t, c1, s1, cache)
        t_dec = (time.perf_counter() - t0) / nstep
        lm = g(t["vocab_size"], H)
        t0 = time.perf_counter(); torch.nn.functional.linear(xe[:, :1], lm); t_lm = time.perf_counter() - t0
    step_s = (n_tiles * v["num_hidden_layers"] * t_vit + t["num_hidden_layers"] * t_pre
              + gen * (t["num_hidden_layers"] * t_dec + t_lm))
    return {"value": gen / step_s, "unit": "tokens/s", "cores": cores, "logical_cpus": ncpu, "kind": "port", "full_depth_run": oracle_full_depth_record(),
            "sample": f"oracle fp32: 1 ViT layer x 1 tile ({t_vit:.2f}s), 1 decoder layer prefill S={S} ({t_pre:.2f}s), "
                      f"{nstep} decode steps x 1 layer at L={S} ({t_dec*1e3:.1f} ms each), lm_head ({t_lm*1e3:.0f} ms); "
                      f"extrapolated to {n_tiles} tiles x {v['num_hidden_layers']} + {t['num_hidden_layers']} layers + {gen} tokens "
                      f"= {step_s:.0f} s/step",
            "decode_tokens_per_sec": 1.0 / (t["num_hidden_layers"] * t_dec + t_lm)}


def _profile_key(path):
    """profiles/rNN_<tag>_... -> (round, len(tag), tag): 'r03_ai' sorts after 'r03_s' (a plain lexicographic sort put it before)."""
    import re
    m = re.match(r"r(\d+)_([a-z]+)_", os.path.basename(path))
    return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, os.path.basename(path))


def pmc_traffic(substrings, which="pmc_traffic", profiles_dir=None, quiet=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (counters cannot be collected together with the timed run).
    The file is the one profiles/MANIFEST.json names under `which` ("pmc_traffic" = configs1, "pmc_traffic_configs2"), written by
    tools/collect_profiles.sh together with the file itself; without a manifest entry: the newest by (round, tag) order.  Picks the
    kernel whose mangled name contains all `substrings`; when the file holds no such kernel (the kernels were renamed after the
    counters were taken) the traffic is None and the source says so -- old counters are never attached to new kernels."""
    import glob
    d = profiles_dir or os.path.join(ROOT, "profiles")
    path = None
    man = os.path.join(d, "MANIFEST.json")
    if os.path.exists(man):
        name = json.load(open(man)).get(which)
        if name:
            path = os.path.join(d, name)
            if not os.path.exists(path):
                print(f"bench.py: profiles/MANIFEST.json names {name} for {which}, which does not exist", file=sys.stderr)
                return None, f"profiles/{name}: named by MANIFEST.json but missing"
    if path is None:
        files = sorted(glob.glob(os.path.join(d, "r*_" + which + ".json")), key=_profile_key)
        if not files:
            return None, None
        path = files[-1]
    ks = json.load(open(path))["kernels"]
    for name, v in ks.items():
        if all(x in name for x in substrings):
            return v["traffic_bytes_per_launch"], "profiles/" + os.path.basename(path)
    if not quiet:
        print(f"bench.py: no kernel matching {substrings} in profiles/{os.path.basename(path)}: PMC traffic not reported", file=sys.stderr)
    return None, f"profiles/{os.path.basename(path)}: no kernel matching {'+'.join(substrings)}"


def roofline_vit_block(total_ms, launches, tiles, max_tiles, rows_per_tile, v_mlp_local, v_hidden):
    """roofline of the ViT fc1 GEMM from its HIP-event brackets.  A pass of `tiles` tiles through a context sized for `max_tiles` runs as
    ceil(tiles / max_tiles) launches per layer (32 tiles at max_tiles 24: 24 + 8), and `total_ms / launches` averages over ALL of them, so
    the flops of one launch are priced at the MEAN tiles per launch (round 4 priced every launch at min(max_tiles, tiles) and reported
    0.68 where 0.45 was true -- VERDICT r04 #9a)."""
    chunks = -(-tiles // max_tiles)
    tiles_per_launch = tiles / chunks
    fl = 2.0 * tiles_per_launch * rows_per_tile * v_mlp_local * v_hidden
    avg_s = total_ms / launches / 1e3
    return {"bound": "mfma", "kernel": "gemm_kernel<EPI_GELU> (ViT fc1)", "achieved": fl / avg_s / 1e12, "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_us": avg_s * 1e6,
            "launches": launches, "flops_per_launch": fl, "tiles_per_launch": tiles_per_launch}


def algorithmic(cfg):
    """algorithmic FLOPs / bytes of the path for a configuration (13B: 11.945 + 0.0498 TF per tile, 13.05 GF per prefill
    token + 200 704 S^2, 14.14 GB of weights + 57 344 B of KV per cached position per decode step)."""
    v, t = cfg.vision, cfg.text
    C, I, Lv = v["hidden_size"], v["intermediate_size"], v["num_hidden_layers"]
    npch = cfg.num_image_tokens
    ntok = npch + 1
    H, It, Lt = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    qd, kvd = t["num_attention_heads"] * t["head_dim"], t["num_key_value_heads"] * t["head_dim"]
    vit_layer = 2.0 * ntok * (4 * C * C + 2 * C * I) + 4.0 * ntok * ntok * C
    vit_tile = Lv * vit_layer + 2.0 * npch * (3 * v["patch_size"] ** 2) * C + 2.0 * npch * (C * H + H * H)
    dec_params = H * (qd + 2 * kvd) + qd * H + 3 * H * It
    lm = t["vocab_size"] * H
    return {"vit_tile": vit_tile,
            "prefill": lambda S: S * 2.0 * Lt * dec_params + Lt * 2.0 * S * S * qd + 2.0 * lm,
            "decode_weight_bytes": (Lt * dec_params + lm) * 2.0,
            "kv_bytes_per_pos": Lt * 2.0 * kvd * 2.0}


def main():
    a = parse()
    if a.workload == "configs4":
        a.no_tp1_check = True          # a quantised computation: there is no TP = 1 16-bit twin to compare ids with
    if a.shard_of and a.gpus > 1:
        raise SystemExit("--shard-of runs on ONE GPU (it is one rank of the group with the exchanges removed)")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    oversub = world > ndev          # debug only (OMCHAT_BENCH_OVERSUBSCRIBE=1): several ranks share a GPU, RCCL cannot run
    torch.cuda.set_device(local_rank % max(ndev, 1))
    if world > 1:
        dist.init_process_group("gloo", init_method="env://")       # bootstrap only; the data path is RCCL + peer kernels inside the library

    from omchat_amd import synth, _lib
    from omchat_amd.config import omchat13b, omchat8b_21, tiny, tiny300m
    from omchat_amd.engine import Engine
    from omchat_amd import tp

    for kv in filter(None, a.tuning.split(",")):
        k, v = kv.split("=")
        os.environ["OMCHAT_ALLOW_TUNING"] = "1"      # measurement hook: this process opts in (include/omchat_hip.h)
        _lib.check(_lib.lib().omchat_op_set_tuning(int(k), int(v)))
    do3 = a.workload == "configs3"
    do4 = a.workload == "configs4"
    if do3:
        cfg = tiny300m() if a.tiny else omchat8b_21()
    else:
        cfg = tiny() if a.tiny else omchat13b()
    n_tiles = a.frames if do4 else (a.images if do3 else 3)
    ntok = cfg.num_image_tokens
    S = n_tiles * ntok + a.text_tokens
    do1 = a.workload in ("both", "configs1", "configs3", "configs4")      # configs3 / 4 run through the batch-1 leg with their own tiles / modes
    do2 = a.workload in ("both", "configs2")
    B2 = a.batch2 if do2 else 1
    shard = a.shard_of if a.shard_of > 1 else 0
    tp_size = shard or world
    full = not a.tiny

    # ---- transports of the tensor-parallel sums
    comm, peer, transport = None, None, {"requested": a.transport}
    if world > 1:
        ok_all = lambda ok: bool(int(_allmin(dist, torch, 1 if ok else 0)))
        if a.transport in ("auto", "rccl") and not oversub:
            try:
                comm = tp.init_comm(rank, world)
                transport["rccl"] = "ok"
            except Exception as e:          # noqa
                transport["rccl"] = f"init failed: {e}"
                comm = None
            if not ok_all(comm is not None):
                comm = None
                transport["rccl"] = transport.get("rccl", "") + " (disabled: not every rank initialised)"
        if a.transport in ("auto", "peer") or comm is None:
            try:
                # ranks sharing ONE GPU (debug): every rank's spinning exchange workgroups sit on the CUs the other ranks' GEMMs need -- with the
                # default 64-workgroup grids eight ranks starved each other into the barrier timeout; 8 workgroups per exchange leave room
                peer = tp.init_peer(rank, world, cap_bytes=64 << 20, max_blocks=8 if oversub else 0)
                ok, detail = tp.peer_selftest(peer, rank, world)
                transport["peer_selftest"] = detail
            except Exception as e:          # noqa
                ok, peer = False, None
                transport["peer_selftest"] = f"failed: {e}"
            if not ok_all(ok):
                peer = None
                transport["peer"] = "disabled (self-test failed on some rank)"
            else:
                transport["peer"] = "ok"
        if comm is None and peer is None:
            raise SystemExit(f"no working tensor-parallel transport: {transport}")
        if comm is not None:
            n = C_int()
            _lib.check(_lib.lib().omchat_comm_count(comm, n.ref()))
            transport["rccl_nranks"] = n.value

    vit_dp = world > 1 and a.vit == "dp"

    def new_engine(cfg_, S_, gen_, b_, tiles_, dtype=None, vision=True):
        e = Engine(cfg_, dtype=dtype or a.dtype, max_seq=S_ + max(gen_, TP1_CHECK_TOKENS + 1) + 8, max_batch=b_, max_tiles=min(24, tiles_),
                   max_prefill_rows=S_ * b_, tp_rank=rank, tp_size=tp_size, comm=comm, vision=vision)
        if peer is not None:
            e.set_peer(peer, 0, all_sizes=(comm is None or a.transport == "peer"))
        if shard:
            e.set_noop_allreduce()
        return e

    eng = new_engine(cfg, S, a.gen, B2, n_tiles * B2, vision=not vit_dp)
    tower = None

    def peer_timed_out():
        """did a barrier spin of the peer exchange give up on ANY rank since the last call?  (the kernels then went on with whatever was in the
        slots: every number after that is void)"""
        if peer is None:
            return 0
        import ctypes as C_
        err = C_.c_int(0)
        _lib.check(_lib.lib().omchat_peer_error(peer, C_.byref(err)))
        return 1 - int(_allmin(dist, torch, 0 if err.value else 1))

    def make_tower():          # replicated vision-only context of this rank (data-parallel tower)
        tw = Engine(cfg, dtype=a.dtype, max_seq=64, max_batch=1, max_tiles=min(24, max(1, -(-n_tiles * B2 // world))), text=False)
        tw.fill_synthetic(0)
        return tw
    if vit_dp:
        tower = make_tower()
    encode = (lambda px: eng.encode_images_dp(tower, px)) if vit_dp else (lambda px: eng.encode_images(px))

    # synthetic inputs, resident in HBM before the timed region
    def make_ids(cfg_, nt, b):
        rows = []
        for i in range(b):
            text = synth.token_ids(a.text_tokens, min(cfg_.text["vocab_size"], 151643), 1 + i).tolist()
            # "<image>\npatch:<image>\npatch:<image>\n{question}" layout (make_context.py:30): sentinel, 1 separator id between
            row = []
            for tix in range(nt):
                row += [-200, text[tix]]
            rows.append(row[:-1] + text[nt - 1:])
        ids = torch.tensor(rows, dtype=torch.int64)
        assert ids.shape[1] == nt + a.text_tokens
        return ids

    def make_inputs(b, e=None, cfg_=None, nt=None):
        e, cfg_, nt = e or eng, cfg_ or cfg, nt or n_tiles
        px = torch.from_numpy(synth.pixels(nt * b, cfg_.vision["image_size"], 0)).to("cuda", e.torch_dtype)
        return px, make_ids(cfg_, nt, b)

    def make_inputs_dynamic(e, cfg_, n_img):
        """configs3: n_img pictures of tile size through dynamic_preprocess + per-tile preprocess on the device (mm_utils.py:276-323;
        a 448 x 448 picture picks the 1 x 1 grid: one tile, no thumbnail) -> one flat tile batch for the whole sample."""
        from omchat_amd.image_processing import HipImageProcessor
        edge = cfg_.vision["image_size"]
        proc = HipImageProcessor(crop_size=edge)
        rng = np.random.default_rng(0)
        pics = [rng.integers(0, 256, (edge, edge, 3), dtype=np.uint8) for _ in range(n_img)]
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tiles = [proc.process_dynamic(p, max_num=6, dtype=e.torch_dtype) for p in pics]
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        px = torch.cat(tiles, 0).contiguous()
        return px, make_ids(cfg_, int(px.shape[0]), 1), sorted(ts[1:])[len(ts[1:]) // 2]

    # ---- TP = N against TP = 1 on rank 0 (same seed, same inputs): greedy ids + full logits.  Default in f16 contexts (round 3): the
    # fp16 comparison is the convincing one -- 5e-3 logit error, every one of the 33 positions clears the margin guard -- and
    # it is independent of the timed run's dtype; --tp1-check-dtype bf16 keeps the round-2 form (8-9 guarded positions of 33).
    tp1_check, tokens_match = None, None
    if world > 1 and not a.no_tp1_check:
        cdt = a.tp1_check_dtype or "f16"
        ref_ids = torch.zeros(TP1_CHECK_TOKENS + 1, dtype=torch.int64)
        ref_logits = None
        if rank == 0:
            e1 = Engine(cfg, dtype=cdt, max_seq=S + TP1_CHECK_TOKENS + 8, max_batch=1, max_tiles=n_tiles, max_prefill_rows=S)
            e1.fill_synthetic(0)
            px, ids = make_inputs(1, e1)
            embeds, lengths, _ = e1.splice(ids, None, e1.encode_images(px))
            logits, _ = e1.prefill(embeds, lengths)
            rows = []
            for i in range(TP1_CHECK_TOKENS + 1):
                rows.append(logits[0].cpu())
                ref_ids[i] = int(torch.argmax(logits[0]))
                if i < TP1_CHECK_TOKENS:
                    _, logits = e1.decode_step(ref_ids[i:i + 1].to(torch.int32), want_logits=True)
            torch.cuda.synchronize()
            ref_logits = torch.stack(rows)
            e1.close(); del e1
            torch.cuda.empty_cache()
        dist.broadcast(ref_ids, src=0)
        tp1 = ref_ids
        # teacher-forced on the TP = 1 ids: the TP = N logits (vocab shards gathered on rank 0) against the TP = 1 logits, and the
        # greedy pick at every position whose TP = 1 top-1 / top-2 margin is above the noise of this very comparison
        same = cdt == a.dtype and not vit_dp
        ec = eng if same else new_engine(cfg, S, TP1_CHECK_TOKENS + 1, 1, n_tiles, dtype=cdt)
        ec.fill_synthetic(0)
        px, ids = make_inputs(1, ec)
        embeds, lengths, _ = ec.splice(ids, None, ec.encode_images(px))
        logits, _ = ec.prefill(embeds, lengths)
        got, shard_rows = [int(ec.argmax(logits)[0])], [logits[0].cpu()]
        for i in range(TP1_CHECK_TOKENS):
            nxt, lg = ec.decode_step(tp1[i:i + 1].to(torch.int32), want_logits=True)
            got.append(int(nxt[0])); shard_rows.append(lg[0].cpu())
        torch.cuda.synchronize()
        if not same:
            ec.close(); del ec
            torch.cuda.empty_cache()
        mine = torch.stack(shard_rows).contiguous()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        if rank == 0:
            Ln, L1 = torch.cat(parts, dim=-1).double(), ref_logits.double()
            rel_err = float((Ln - L1).norm() / L1.norm())
            rms = float((Ln - L1).pow(2).mean().sqrt())
            top2 = torch.topk(ref_logits, 2, dim=-1).values
            margin = (top2[:, 0] - top2[:, 1]).double()
            guard = max(MARGIN_GUARD[cdt], 6.0 * rms)      # a gap difference has sd sqrt(2) x rms: ~4 sd
            want = [int(x) for x in tp1]
            guarded = [i for i in range(len(want)) if float(margin[i]) > guard]
            # two 16-bit evaluation orders of the full 45 + 28 layer model: measured 4.2e-2 (bf16) / 5.3e-3 (f16) for TP = 2 and TP = 4 alike,
            # next to 3.7e-2 between the prefill and the decode kernels of ONE bf16 context (tests/test_gpu_fullsize.py CONSIST_TOL); a
            # sharding or transport error shows as >= 1e-1 (the two-shot segment bug of round 2 read 2.3e-1).  The tiny geometry keeps TOL_DEEP.
            tol = ((6e-2 if cdt == "bf16" else 1.2e-2) if not a.tiny else (3e-2 if cdt == "bf16" else 6e-3))
            tp1_check = {"mode": "teacher-forced on the TP=1 ids", "dtype": cdt, "compared": len(want),
                         "equal": sum(int(g == w) for g, w in zip(got, want)),
                         "guarded": len(guarded), "guarded_equal": sum(int(got[i] == want[i]) for i in guarded), "margin_guard": guard,
                         "logit_rel_err": rel_err, "logit_rms_diff": rms, "logit_tolerance": tol, "min_margin": float(margin.min()),
                         "median_margin": float(margin.median())}
            tokens_match = bool(tp1_check["guarded_equal"] == tp1_check["guarded"] and rel_err < tol)
        if peer_timed_out():
            raise SystemExit("tp1_check: a peer-exchange barrier timed out on some rank (ranks oversubscribing one GPU starve each other; on one "
                             "rank per GPU this means a lost peer): the comparison is void")

    eng.fill_synthetic(0, local=bool(shard))
    if a.graph and world == 1 and not shard:
        eng.enable_decode_graph(True)      # one graph launch per token; every 8th step stays eager for the HIP-event brackets
    if do4:
        eng.enable_fp8_decode(True); eng.enable_fp8_kv(True); eng.enable_fp8_prefill(True)

    ev = lambda: torch.cuda.Event(enable_timing=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(e, enc, px, ids, steps, warmup, gen):
        b = ids.shape[0]

        def step():
            t = [ev() for _ in range(5)]
            t[0].record()
            feats = enc(px)
            t[1].record()
            embeds, lengths, _ = e.splice(ids, None, feats)
            logits, _ = e.prefill(embeds, lengths)
            tok = e.argmax(logits)
            t[2].record()
            first = tok.clone()
            t[3].record()
            out = [first]
            for _ in range(gen - 1):
                tok, _ = e.decode_step(tok)
                out.append(tok)
            t[4].record()
            torch.cuda.synchronize()
            return (t[0].elapsed_time(t[1]), t[1].elapsed_time(t[2]), t[3].elapsed_time(t[4]), torch.stack(out))

        for _ in range(warmup):
            step()
        e.prof_enable(True)
        for c in range(3):
            e.prof_read(c, reset=True)
        barrier()
        t0 = time.perf_counter()
        parts = [step() for _ in range(steps)]
        barrier()
        wall = time.perf_counter() - t0
        e.prof_enable(False)
        if world > 1:
            tw = torch.tensor([wall], dtype=torch.float64)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            wall = float(tw[0])
        prof = {c: e.prof_read(c) for c in range(3)}
        med = lambda xs: sorted(xs)[len(xs) // 2]
        tiles = int(px.shape[0])
        return dict(wall=wall, steps=steps, b=b, gen=gen, tiles=tiles, S=int(ids.shape[1]) + (tiles // b) * (e.ntok - 1),
                    prof=prof, vit_ms=med([p[0] for p in parts]), pre_ms=med([p[1] for p in parts]), dec_ms=med([p[2] for p in parts]),
                    max_tiles=e.c.max_tiles, local=e.local)

    def summarise(r, cfg_, decode_bytes=None):
        """decode_bytes(b, L) -> algorithmic bytes of one decode step (default: 16-bit weights + 16-bit KV)"""
        al = algorithmic(cfg_)
        b, gen, S_ = r["b"], r["gen"], r["S"]
        tiles = r["tiles"]
        dec_step_s = r["dec_ms"] / 1e3 / max(gen - 1, 1)
        vit_s, pre_s = r["vit_ms"] / 1e3, r["pre_ms"] / 1e3
        db = decode_bytes or (lambda bb, L: al["decode_weight_bytes"] + al["kv_bytes_per_pos"] * L * bb)
        return {
            "tokens_per_sec": b * gen * r["steps"] / r["wall"], "samples_per_sec": b * r["steps"] / r["wall"], "ms_per_step": r["wall"] / r["steps"] * 1e3,
            "steps": r["steps"], "batch": b, "decode_tokens_per_sec": b / dec_step_s, "images_per_sec": tiles / vit_s,
            "ttft_ms_p50": r["vit_ms"] + r["pre_ms"], "vit_ms_p50": r["vit_ms"], "prefill_ms_p50": r["pre_ms"],
            "decode_ms_per_step_p50": dec_step_s * 1e3,
            "vit_mfma_frac": tiles * al["vit_tile"] / vit_s / 1e12 / MFMA_PEAK_TFLOPS / tp_size if full else None,
            # the north star states its 40 % target on "ViT + decoder prefill": both together over the time to first token
            "ttft_mfma_frac": (tiles * al["vit_tile"] + b * al["prefill"](S_)) / (vit_s + pre_s) / 1e12 / MFMA_PEAK_TFLOPS / tp_size if full else None,
            "prefill_mfma_frac": b * al["prefill"](S_) / pre_s / 1e12 / MFMA_PEAK_TFLOPS / tp_size if full else None,
            # algorithmic bytes of one decode step: weights once + KV of every sequence at the mean decode length
            "decode_hbm_frac": (db(b, S_ + gen / 2) / tp_size / dec_step_s / 1e9 / HBM_PEAK_GBS) if full else None,
        }

    def rooflines(r, cfg_, f8=False):
        b, S_ = r["b"], r["S"]
        v, t = cfg_.vision, cfg_.text
        prof, ld = r["prof"], r["local"]
        gu_bytes = 2.0 * ld["t_mlp"] * t["hidden_size"] * (1 if f8 else 2)      # algorithmic bytes per launch = the (rank-local) gate|up weights
        ms, n = prof[_lib.PROF_DECODE_GATEUP]
        roof = roof_pre = roof_vit = None
        plain = tp_size == 1 and full and cfg_.vision["hidden_size"] == 3200
        if n:
            avg_s = ms / n / 1e3
            if f8:
                kern = "gemv_rows_kernel<EPI_SWIGLU, F8> (decode gate|up e4m3 weight stream, batch 1)"
            elif b == 1:
                kern = "gemv_rows_norm_kernel<EPI_SWIGLU, 1 pair per wave> (decode post-attention RMSNorm + gate|up weight stream, batch 1)"
            else:
                kern = f"gemv_xs_kernel<EPI_SWIGLU, NB={2 if b > 16 else 1}> (decode gate|up weight stream, x-stationary, batch {b})"
            tr, src = (None, None)
            if plain and not f8:
                # symbol patterns: ONE table shared with tools/roofline_table.py and checked against the built library by a CPU test
                from tools import kernel_roles
                if b == 1:      # the launch that streams gate|up in the default configuration first, then the forms behind tuning keys 38 / 16 / 14
                    subs = kernel_roles.decode_gate_up_b1()
                    for i, sub in enumerate(subs):
                        tr, src = pmc_traffic(sub, quiet=i + 1 < len(subs))
                        if tr is not None:
                            break
                else:
                    tr, src = pmc_traffic(kernel_roles.decode_gate_up_batched(b)[0], "pmc_traffic_configs2")
            roof = {"bound": "hbm", "kernel": kern, "achieved": gu_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gu_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src, "avg_launch_us": avg_s * 1e6,
                    "launches": n, "bytes_per_launch": gu_bytes}
        ms, n = prof[_lib.PROF_PREFILL_GATEUP]
        if n:
            fl = 2.0 * b * S_ * (2 * ld["t_mlp"]) * t["hidden_size"]
            avg_s = ms / n / 1e3
            from tools import kernel_roles
            tr, src = pmc_traffic(kernel_roles.prefill_gate_up()[0]) if (plain and b == 1 and not f8) else (None, None)
            peak = MFMA_PEAK_FP8_TFLOPS if f8 else MFMA_PEAK_TFLOPS      # e4m3 operands are priced against the fp8 peak
            roof_pre = {"bound": "mfma", "kernel": "gemm8_kernel<256x256,EPI_SWIGLU" + (",F8" if f8 else "") + "> (prefill gate|up)", "achieved": fl / avg_s / 1e12,
                        "peak": peak, "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / peak, "traffic": tr,
                        "traffic_source": src, "avg_launch_us": avg_s * 1e6, "launches": n, "flops_per_launch": fl}
            if f8:
                roof_pre["frac_of_bf16_peak"] = fl / avg_s / 1e12 / MFMA_PEAK_TFLOPS
        ms, n = prof[_lib.PROF_VIT_FC1]
        if n:
            roof_vit = roofline_vit_block(ms, n, r["tiles"], r["max_tiles"], cfg_.num_image_tokens + 1, ld["v_mlp"], v["hidden_size"])
        return roof, roof_pre, roof_vit

    frontend3_ms = None
    r1 = None
    if do3:
        px1, ids1, frontend3_ms = make_inputs_dynamic(eng, cfg, n_tiles)
    elif do1:
        px1, ids1 = make_inputs(1)
    if do1:
        r1 = run_workload(eng, encode, px1, ids1, a.steps, a.warmup, a.gen)
    r2 = None
    if do2:
        steps2 = a.steps if not do1 else (a.steps2 or min(a.steps, 3))
        px2, ids2 = make_inputs(B2)
        r2 = run_workload(eng, encode, px2, ids2, steps2, a.warmup if not do1 else 1, a.gen)
        del px2
    comm_stats = eng.comm_stats() if world > 1 else None
    peer_timeouts = peer_timed_out() if world > 1 else None
    if peer_timeouts:
        raise SystemExit("bench.py: a peer-exchange barrier timed out on some rank inside the measured steps: the kernels went on with stale slots, "
                         "the numbers are void (ranks oversubscribing one GPU can starve each other; one rank per GPU: a lost peer)")
    # data-parallel tower beside the tensor-parallel headline: same tiles, replicated tower, one gather (all ranks take part)
    vit_dp_side = None
    if world > 1 and a.vit == "both":
        tower = make_tower()
        vit_dp_side = {}
        for nm, bb in (("configs1", 1),) + ((("configs2", B2),) if do2 else ()):
            px, _ = make_inputs(bb)
            eng.encode_images_dp(tower, px)
            ts = []
            for _ in range(3):
                barrier(); e0, e1 = ev(), ev(); e0.record()
                f_dp = eng.encode_images_dp(tower, px)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            f_tp = eng.encode_images(px); torch.cuda.synchronize()
            d = (f_dp.float() - f_tp.float()).norm() / f_tp.float().norm()
            vit_dp_side[nm] = {"tiles": n_tiles * bb, "vit_ms_p50": sorted(ts)[1], "rel_diff_vs_tp_tower": float(d)}
        tower.close()
    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    al4 = algorithmic(cfg)
    bytes_f8 = lambda bb, L: al4["decode_weight_bytes"] / 2 + al4["kv_bytes_per_pos"] / 2 * L * bb
    head = r1 if do1 else r2
    hs = summarise(head, cfg, bytes_f8 if do4 else None)
    roof, roof_pre, roof_vit = rooflines(head, cfg, f8=do4)
    if a.tiny:
        name = "TINY DEBUG GEOMETRY"
    elif do3:
        name = "OmChat-2.1-8B (InternViT-300M 24L + Qwen2-7B 28L)"
    else:
        name = "OmChat-13B (InternViT-6B 45L + Qwen2-7B 28L)"
    wl1 = (f"{name}, configs[1]: 1 sample = {n_tiles} tiles of 448x448 + {a.text_tokens} text ids -> prefill S={S}, "
           f"{a.gen} greedy decode tokens, batch 1")
    wl3 = (f"{name}, configs[3]: 1 sample = {n_tiles} pictures of 448x448 -> dynamic tiling -> {n_tiles} tiles in one tower batch + {a.text_tokens} "
           f"text ids -> prefill S={S}, {a.gen} greedy decode tokens, batch 1")
    wl4 = (f"{name}, configs[4]: one {n_tiles}-frame clip = {n_tiles} tiles + {a.text_tokens} text ids -> prefill S={S}, {a.gen} greedy decode "
           f"tokens, batch 1; fp8: e4m3 decode weights + e4m3 KV cache + fp8 x fp8 MFMA qkv / gate|up prefill GEMMs")
    wl2 = (f"{name}, configs[2]: {B2} samples per step = {n_tiles * B2} tiles + {B2} x {a.text_tokens} text ids -> batch prefill {B2} x {S}, "
           f"{a.gen} batched greedy decode steps")
    wl = wl4 if do4 else (wl3 if do3 else (wl1 if do1 else wl2))
    res = {
        "metric": "images/sec prefill + decode tokens/sec, OmChat-13B TP=1/8; p50 TTFT",
        "value": hs["tokens_per_sec"], "unit": "tokens/s", "n_gpus": world, "steps": head["steps"], "warmup": a.warmup,
        "ms_per_step": hs["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": wl, "parallelism": f"tp{world}", "tiles": n_tiles * head["b"], "prefill_tokens": S * head["b"],
                   "gen_tokens": a.gen, "batch": head["b"]},
        "decode_tokens_per_sec": hs["decode_tokens_per_sec"], "images_per_sec": hs["images_per_sec"], "ttft_ms_p50": hs["ttft_ms_p50"],
        "vit_ms_p50": hs["vit_ms_p50"], "prefill_ms_p50": hs["prefill_ms_p50"], "decode_ms_per_token_p50": hs["decode_ms_per_step_p50"],
        "vit_mfma_frac": hs["vit_mfma_frac"], "prefill_mfma_frac": hs["prefill_mfma_frac"], "ttft_mfma_frac": hs["ttft_mfma_frac"],
        "decode_hbm_frac": hs["decode_hbm_frac"],
        "roofline": roof, "roofline_prefill": roof_pre, "roofline_vit": roof_vit,
        "device_gb": eng.device_bytes() / 1e9,
        "decode_graph": eng.decode_graph_stats() if (a.graph and world == 1 and not shard) else None,
    }
    if shard:
        res["shard_of"] = shard
        res["comm_stats"] = eng.comm_stats()      # sp_reduce_scatters > 0: the sequence-parallel form ran (tools/tp_projection.py prices it)
        res["config"]["parallelism"] = f"rank 0 of tp{shard}, exchanges removed (omchat_allreduce_noop)"
        res["shard_note"] = ("ONE rank's share of a TP group on one GPU: shard shapes and launch sequence, no communication; every *_frac is "
                             "this rank's algorithmic share (1 / N of the FLOPs / bytes) over its own time.  Not a throughput claim.")
    if do3:
        res["frontend_ms_p50"] = frontend3_ms
    if do4 and full:
        res["dtype"] = a.dtype + " activations, e4m3 weights (decode GEMVs, qkv / gate|up prefill GEMMs) and KV cache"
        res["decode_hbm_note"] = "algorithmic bytes per token: 7.07 GB of e4m3 weights + 28 672 B of e4m3 KV per cached position"
    if world > 1:
        res["rccl_nranks"] = transport.get("rccl_nranks")
        res["tokens_match_tp1"] = tokens_match
        res["tp1_check"] = tp1_check
        res["transport"] = transport
        res["comm_stats"] = comm_stats
        res["peer_timeouts"] = peer_timeouts
        res["config"]["parallelism"] = f"tp{world}" + (" (vision tower data-parallel over tiles)" if vit_dp else "")
        if vit_dp_side is not None:
            res["vit_data_parallel"] = dict(vit_dp_side, note="replicated tower, tiles dealt to the ranks, one all-reduce gathers the features; NOT part of `value`")
    if do1 and do2:
        s2 = summarise(r2, cfg)
        ro2, rp2, rv2 = rooflines(r2, cfg)
        s2.update({"workload": wl2, "roofline": ro2, "roofline_prefill": rp2, "roofline_vit": rv2,
                   "note": "side measurement in the same process and context; NOT part of `value`"})
        res["configs2"] = s2
    # weight-only fp8 decode (row f-2 / configs[4]), outside the timed region: same prompt, 64 greedy tokens on the e4m3 replica
    if world == 1 and not shard and not a.no_fp8 and not do4 and not do3:
        px, ids = make_inputs(1)
        eng.enable_fp8_decode(True)
        n8 = min(64, a.gen)
        t8 = []
        for _ in range(2):
            feats = eng.encode_images(px)
            embeds, lengths, _ = eng.splice(ids, None, feats)
            logits, _ = eng.prefill(embeds, lengths)
            tok = eng.argmax(logits)
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(n8):
                tok, _ = eng.decode_step(tok)
            e1.record(); torch.cuda.synchronize()
            t8.append(e0.elapsed_time(e1) / n8)
        eng.enable_fp8_decode(False)
        res["fp8_decode"] = {"decode_ms_per_token": min(t8), "decode_tokens_per_sec": 1e3 / min(t8),
                             "hbm_frac": ((al4["decode_weight_bytes"] / 2 + al4["kv_bytes_per_pos"] * S) / (min(t8) / 1e3) / 1e9 / HBM_PEAK_GBS) if full else None,
                             "note": "decoder GEMV weights as OCP e4m3 + per-row fp32 scale (7.07 GB/step instead of 14.14); "
                                     "prefill, KV cache and activations stay 16-bit; NOT part of `value`"}
    # the drop-in entry (single_inference.py:53-62): model.generate(input_ids, images=...) on the same sample -- tower, splice, prefill and
    # the greedy loop with the host looking at every token (EOS / streamer), next to the engine-level loop of `value`
    if world == 1 and not shard and do1 and not do3 and not do4:
        from omchat_amd.model.omchat_qwen2 import OmChatQwen2ForCausalLM
        model = OmChatQwen2ForCausalLM(cfg.clone(), eng)
        px, ids = make_inputs(1)
        model.generate(ids, images=px, max_new_tokens=8, eos_token_id=None)      # warm
        tg, t1 = [], []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.generate(ids, images=px, max_new_tokens=1, eos_token_id=None)
            torch.cuda.synchronize(); t1.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            out_ids = model.generate(ids, images=px, max_new_tokens=a.gen, eos_token_id=None)
            torch.cuda.synchronize(); tg.append(time.perf_counter() - t0)
        assert out_ids.shape[1] == ids.shape[1] + a.gen
        whole, first = sorted(tg)[1], sorted(t1)[1]
        res["generate"] = {"tokens_per_sec": a.gen / whole, "ms_per_call": whole * 1e3, "first_token_ms": first * 1e3,
                           "decode_tokens_per_sec": (a.gen - 1) / (whole - first) if a.gen > 1 else None,
                           "vs_engine_loop": (a.gen / whole) / hs["tokens_per_sec"],
                           "note": "OmChatQwen2ForCausalLM.generate(input_ids, images=...) wall time per call (host-side splice plan, one pinned "
                                   "token copy + event wait per generated token, step k + 1 enqueued before token k is read); NOT `value`"}
        res["generate_tokens_per_sec"] = a.gen / whole
        if B2 >= 2 and n_tiles >= 2:
            # a PADDED batch through the same entry: two rows of different spliced length (n_tiles tiles vs 1 tile + the same text), the
            # shorter one right-padded -> the reference's decode branch (omchat_arch.py:61-70: common cache slot, sum(mask) - 1 positions, token-level
            # key mask), which the engine runs from device-resident masks / positions (omchat_decode_step_masked_next: no per-step synchronisation)
            ids_a = make_ids(cfg, n_tiles, 1)[0].tolist()
            ids_b = make_ids(cfg, 1, 1)[0].tolist()
            T2 = max(len(ids_a), len(ids_b))
            ids2 = torch.zeros(2, T2, dtype=torch.int64); mask2 = torch.zeros(2, T2, dtype=torch.int64)
            for i, r in enumerate((ids_a, ids_b)):
                ids2[i, :len(r)] = torch.tensor(r); mask2[i, :len(r)] = 1
            px2, _ = make_inputs(1, nt=n_tiles + 1)
            model.generate(ids2, images=px2, attention_mask=mask2, max_new_tokens=8, eos_token_id=None, pad_token_id=0)      # warm
            tp, tp1 = [], []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                model.generate(ids2, images=px2, attention_mask=mask2, max_new_tokens=1, eos_token_id=None, pad_token_id=0)
                torch.cuda.synchronize(); tp1.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                o2 = model.generate(ids2, images=px2, attention_mask=mask2, max_new_tokens=a.gen, eos_token_id=None, pad_token_id=0)
                torch.cuda.synchronize(); tp.append(time.perf_counter() - t0)
            assert o2.shape == (2, T2 + a.gen) and getattr(model, "_padded_batch", False)
            wp, fp = sorted(tp)[1], sorted(tp1)[1]
            res["generate_padded_batch"] = {"rows": 2, "spliced_lengths": [n_tiles * ntok + a.text_tokens, ntok + a.text_tokens], "tokens_per_sec": 2 * a.gen / wp,
                                            "decode_ms_per_step": (wp - fp) / max(a.gen - 1, 1) * 1e3, "ms_per_call": wp * 1e3,
                                            "note": "generate() on a ragged right-padded batch: masked decode steps as the reference computes them "
                                                    "(omchat_arch.py:61-70), mask and positions resident on the device, step k + 1 enqueued before token k is read"}
    if world == 1 and not shard:
        n_f, bits = eng.fused_status()
        res["fused_decode"] = {"launches": n_f, "timeout_bits": bits,
                               "note": "attention + merge + o_proj of a batch-1 decode step as one launch with in-launch hand-offs (csrc/experiments/fused_decode.hip)"}
        if bits:
            raise SystemExit(f"bench.py: a hand-off of the fused decode launch timed out (bits {bits:#x}): results invalid")
    rgb = pins = None
    if world == 1 and not shard and not do3:
        # image front-end (row f-1), outside the timed region: raw RGB bytes on the host -> normalised tiles in HBM
        pins = [(448, 896), (896, 448), (896, 896), (1344, 448), (448, 1344), (1344, 1344)]
        rgb = np.random.default_rng(0).integers(0, 256, (380, 570, 3), dtype=np.uint8)       # size of the reference's sample picture -> 3 tiles
        from omchat_amd.image_processing import HipImageProcessor
        proc = HipImageProcessor(crop_size=cfg.vision["image_size"])
        fe = []
        for _ in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            proc.process_anyres(rgb, pins, dtype=eng.torch_dtype)
            torch.cuda.synchronize(); fe.append((time.perf_counter() - t0) * 1e3)
        res["frontend_ms_p50"] = sorted(fe[2:])[len(fe[2:]) // 2]

    # ---- side blocks of the default run (N = 1): BASELINE configs[3] and configs[4], each in its own context
    side = a.workload == "both" and world == 1 and not shard and not a.no_side and full
    if side:
        eng.close()
        del eng
        torch.cuda.empty_cache()
        # configs[3]
        cfg3 = omchat8b_21()
        S3 = a.images * cfg3.num_image_tokens + a.text_tokens
        e3 = new_engine(cfg3, S3, a.gen, 1, a.images)
        e3.fill_synthetic(0)
        px3, ids3, fe3 = make_inputs_dynamic(e3, cfg3, a.images)
        r3 = run_workload(e3, e3.encode_images, px3, ids3, min(a.steps, 3), 1, a.gen)
        s3 = summarise(r3, cfg3)
        ro, rp, rv = rooflines(r3, cfg3)
        s3.update({"workload": f"OmChat-2.1-8B (InternViT-300M 24L + Qwen2-7B 28L), configs[3]: 1 sample = {a.images} pictures of 448x448 -> dynamic "
                               f"tiling -> {int(px3.shape[0])} tiles in one tower batch + {a.text_tokens} text ids -> prefill S={r3['S']}, {a.gen} greedy "
                               f"decode tokens, batch 1",
                   "roofline": ro, "roofline_prefill": rp, "roofline_vit": rv, "frontend_ms_p50": fe3, "device_gb": e3.device_bytes() / 1e9,
                   "note": "side measurement in its own context; NOT part of `value`"})
        res["configs3"] = s3
        e3.close(); del e3, px3
        torch.cuda.empty_cache()
        # configs[4]
        gen4 = min(64, a.gen)
        S4 = a.frames * cfg.num_image_tokens + a.text_tokens
        e4 = new_engine(cfg, S4, gen4, 1, a.frames)
        e4.fill_synthetic(0)
        e4.enable_fp8_decode(True); e4.enable_fp8_kv(True); e4.enable_fp8_prefill(True)
        px4, ids4 = make_inputs(1, e4, cfg, a.frames)
        r4 = run_workload(e4, e4.encode_images, px4, ids4, min(a.steps, 2), 1, gen4)
        s4 = summarise(r4, cfg, bytes_f8)
        ro, rp, rv = rooflines(r4, cfg, f8=True)
        s4.update({"workload": f"OmChat-13B, configs[4]: one {a.frames}-frame clip = {a.frames} tiles + {a.text_tokens} text ids -> prefill S={S4}, {gen4} "
                               f"greedy decode tokens, batch 1; fp8: e4m3 decode weights + e4m3 KV cache + fp8 x fp8 MFMA qkv / gate|up prefill GEMMs",
                   "dtype": a.dtype + " activations, e4m3 weights (decode GEMVs, qkv / gate|up prefill GEMMs) and KV cache",
                   "roofline": ro, "roofline_prefill": rp, "roofline_vit": rv, "device_gb": e4.device_bytes() / 1e9,
                   "note": "side measurement in its own context, a quantised computation; NOT part of `value`"})
        res["configs4"] = s4
        e4.close(); del e4, px4
        torch.cuda.empty_cache()
    # GEMM tile choices come from the committed omchat_amd/gemm_tune_gfx950.txt: 0 = no first-use tuning ran in this process
    res["gemm_tune_measurements"] = int(_lib.lib().omchat_gemm_tune_runs())
    if world == 1 and not shard and not a.no_cpu_baseline:          # CPU baseline: rank 0 at N = 1 only
        res["cpu_baseline"] = cpu_baseline(cfg, S, a.gen, n_tiles)
        if rgb is not None:
            from PIL import Image
            from transformers import CLIPImageProcessor
            from oracle.preproc import pil_process_anyres_image      # the reference's PIL recipe (checker side), timed as the CPU baseline
            cp = CLIPImageProcessor(crop_size=448, do_center_crop=True, do_normalize=True, do_resize=True,
                                    image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225], size=448)
            img = Image.fromarray(rgb)
            pil_process_anyres_image(img, cp, pins)
            t0 = time.perf_counter()
            for _ in range(3):
                pil_process_anyres_image(img, cp, pins)
            res["cpu_baseline"]["frontend_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()


class C_int:
    """tiny ctypes int holder (keeps ctypes out of the module namespace of the spawning parent)"""
    def __init__(self):
        import ctypes
        self._c = ctypes.c_int(0)
        self._ctypes = ctypes

    def ref(self):
        return self._ctypes.byref(self._c)

    @property
    def value(self):
        return self._c.value


def _allmin(dist, torch, v):
    t = torch.tensor([v], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t[0])


if __name__ == "__main__":
    main()
