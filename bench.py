#!/usr/bin/env python3
"""bench.py -- OmChat-13B hot path on MI355X: ViT (3 tiles) -> projector -> splice -> prefill (S = 3584) -> greedy decode.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gen G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one sample of BASELINE.json configs[1]: one 448x448 picture => 3 anyres tiles (thumbnail + 2, because
select_best_resolution((448,448)) = (448,896), mm_utils.py:28-37,151) + 512 text ids => S = 3*1024 + 512 = 3584 prefill
tokens, then G greedy decode tokens (EOS disabled).  Inputs are resident in HBM before the timed region.  Weights:
deterministic synthetic (omchat_amd/synth.py) at the full OmChat-13B geometry, generated on the device.

Prints ONE JSON line (rank 0).  `value` = generated tokens / second over whole steps (prefill included);
decode-only tokens/s, ViT tiles/s and p50 TTFT are reported beside it, with the roofline of the dominant kernel
(decode gate|up weight-streaming GEMV, HBM-bound) and of the dominant prefill kernel (gate|up MFMA GEMM), both from
HIP events recorded on the launch stream inside the timed region, and a CPU baseline (the oracle on a bounded sample).
"""
import argparse
import json
import numpy as np
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/f16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="replay each decode step as one captured hipGraph instead of ~230 eager launches "
                    "(measured SLOWER on ROCm 7.2 / MI355X: 3.21 vs 2.96 ms per token, so it is off by default)")
    ap.add_argument("--no-fp8", action="store_true", help="skip the (untimed) weight-only fp8 decode measurement")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gen", type=int, default=256, help="greedy decode tokens per step")
    ap.add_argument("--text-tokens", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny geometry (NOT the benchmark config)")
    return ap.parse_args()


def cpu_baseline(cfg, S, gen, n_tiles):
    """The oracle (kind 'port') on this host's cores, bounded sample: 1 ViT layer on 1 tile (1025 tokens), 1 decoder
    layer prefill at S, 4 decode steps of 1 decoder layer at L = S, lm_head once; extrapolated to the whole step."""
    import torch
    import oracle
    from oracle.decoder import qwen2_layer, rope_cos_sin
    torch.manual_seed(0)
    # thread count: the best of a few candidates on a decode-shaped matvec (all 256 SMT threads of the GPU host is pathological)
    ncpu = os.cpu_count() or 1
    wprobe, xprobe = torch.randn(18944, 3584), torch.randn(1, 3584)
    best = (1e9, 1)
    for nthr in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nthr)
        torch.nn.functional.linear(xprobe, wprobe)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(xprobe, wprobe)
        best = min(best, (time.perf_counter() - t0, nthr))
    cores = best[1]
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
    return {"value": gen / step_s, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32: 1 ViT layer x 1 tile ({t_vit:.2f}s), 1 decoder layer prefill S={S} ({t_pre:.2f}s), "
                      f"{nstep} decode steps x 1 layer at L={S} ({t_dec*1e3:.1f} ms each), lm_head ({t_lm*1e3:.0f} ms); "
                      f"extrapolated to {n_tiles} tiles x {v['num_hidden_layers']} + {t['num_hidden_layers']} layers + {gen} tokens "
                      f"= {step_s:.0f} s/step",
            "decode_tokens_per_sec": 1.0 / (t["num_hidden_layers"] * t_dec + t_lm)}


def pmc_traffic(substrings):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json; counters cannot be
    collected together with the timed run).  Picks the kernel whose mangled name contains all `substrings`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    ks = json.load(open(files[-1]))["kernels"]
    for name, v in ks.items():
        if all(x in name for x in substrings):
            return v["traffic_bytes_per_launch"]
    return None


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("gloo", init_method="env://")       # bootstrap only; the data path uses RCCL inside the library

    from omchat_amd import synth, _lib
    from omchat_amd.config import omchat13b, tiny
    from omchat_amd.engine import Engine
    from omchat_amd.tp import init_comm

    cfg = tiny() if a.tiny else omchat13b()
    n_tiles = 3
    ntok = cfg.num_image_tokens
    S = n_tiles * ntok + a.text_tokens
    comm = init_comm(rank, world) if world > 1 else None
    eng = Engine(cfg, dtype=a.dtype, max_seq=S + a.gen + 8, max_batch=1, max_tiles=n_tiles, max_prefill_rows=S,
                 tp_rank=rank, tp_size=world, comm=comm)
    eng.fill_synthetic(0)
    if a.graph and world == 1:
        eng.enable_decode_graph(True)      # one graph launch per token; every 8th step stays eager for the HIP-event brackets

    # synthetic inputs, resident in HBM before the timed region
    px = torch.from_numpy(synth.pixels(n_tiles, cfg.vision["image_size"], 0)).to("cuda", eng.torch_dtype)
    text = synth.token_ids(a.text_tokens, min(cfg.text["vocab_size"], 151643), 1).tolist()
    # "<image>\npatch:<image>\npatch:<image>\n{question}" layout (make_context.py:30): sentinel, 1 separator id between
    ids = [-200, text[0], -200, text[1], -200] + text[2:]
    ids = torch.tensor([ids], dtype=torch.int64)
    assert ids.shape[1] - n_tiles + n_tiles * ntok == S

    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(timed):
        e = [ev() for _ in range(5)]
        e[0].record()
        feats = eng.encode_images(px)
        e[1].record()
        embeds, lengths, _ = eng.splice(ids, None, feats)
        logits, _ = eng.prefill(embeds, lengths)
        tok = eng.argmax(logits)
        e[2].record()
        first = tok.clone()
        e[3].record()
        out = [first]
        for _ in range(a.gen - 1):
            tok, _ = eng.decode_step(tok)
            out.append(tok)
        e[4].record()
        torch.cuda.synchronize()
        return (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[3].elapsed_time(e[4]), torch.stack(out).view(-1))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    eng.prof_enable(True)
    for c in range(3):
        eng.prof_read(c, reset=True)
    barrier()
    t0 = time.perf_counter()
    parts = []
    for _ in range(a.steps):
        parts.append(step(True))
    barrier()
    wall = time.perf_counter() - t0
    eng.prof_enable(False)
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw[0])

    prof = {c: eng.prof_read(c) for c in range(3)}
    if rank != 0:
        return
    vit_ms = sorted(p[0] for p in parts); pre_ms = sorted(p[1] for p in parts); dec_ms = sorted(p[2] for p in parts)
    med = lambda xs: xs[len(xs) // 2]
    v, t = cfg.vision, cfg.text
    ld = eng.local
    # dominant kernel of the step by time: decode gate|up GEMV.  Algorithmic bytes per launch = its (rank-local) weights.
    gu_bytes = 2.0 * ld["t_mlp"] * t["hidden_size"] * 2
    ms, n = prof[_lib.PROF_DECODE_GATEUP]
    roof = None
    if n:
        avg_s = ms / n / 1e3
        roof = {"bound": "hbm", "kernel": "gemv_rows_kernel<EPI_SWIGLU> (decode gate|up weight stream)", "achieved": gu_bytes / avg_s / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gu_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic(["gemv_rows_kernelI", "Li4ELi4ELi4E"]) if (world == 1 and not a.tiny) else None,     # <T, EPI_SWIGLU=4, RR=4, WAVES=4>
                "avg_launch_us": avg_s * 1e6, "launches": n, "bytes_per_launch": gu_bytes}
    ms, n = prof[_lib.PROF_PREFILL_GATEUP]
    roof_pre = None
    if n:
        fl = 2.0 * S * (2 * ld["t_mlp"]) * t["hidden_size"]
        avg_s = ms / n / 1e3
        roof_pre = {"bound": "mfma", "kernel": "gemm_kernel<256x256,EPI_SWIGLU> (prefill gate|up)", "achieved": fl / avg_s / 1e12,
                    "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / MFMA_PEAK_TFLOPS,
                    "traffic": pmc_traffic(["gemm8_kernel", "Li4E"]) if (world == 1 and not a.tiny) else None,
                    "avg_launch_us": avg_s * 1e6, "launches": n, "flops_per_launch": fl}
    ms, n = prof[_lib.PROF_VIT_FC1]
    roof_vit = None
    if n:
        fl = 2.0 * n_tiles * (ntok + 1) * ld["v_mlp"] * v["hidden_size"]
        avg_s = ms / n / 1e3
        roof_vit = {"bound": "mfma", "kernel": "gemm_kernel<EPI_GELU> (ViT fc1)", "achieved": fl / avg_s / 1e12, "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_us": avg_s * 1e6,
                    "launches": n, "flops_per_launch": fl}
    vit_flops = n_tiles * (45 * (2 * 1025 * 122.88e6 + 4 * 1025 ** 2 * 3200) + 2 * 1024 * 588 * 3200 + 2 * 1024 * (3200 * 3584 + 3584 ** 2)) \
        if not a.tiny else 0.0
    pre_flops = (S * 2 * 28 * 233.06e6 + 28 * 2 * S * S * 3584 + 2 * 545e6) if not a.tiny else 0.0
    res = {
        "metric": "images/sec prefill + decode tokens/sec, OmChat-13B TP=1/8; p50 TTFT",
        "value": a.gen * a.steps / wall, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": ("TINY DEBUG GEOMETRY" if a.tiny else "OmChat-13B (InternViT-6B 45L + Qwen2-7B 28L)") +
                   f", configs[1]: 1 sample = {n_tiles} tiles of 448x448 + {a.text_tokens} text ids -> prefill S={S}, "
                   f"{a.gen} greedy decode tokens, batch 1", "parallelism": f"tp{world}", "tiles": n_tiles, "prefill_tokens": S,
                   "gen_tokens": a.gen},
        "decode_tokens_per_sec": (a.gen - 1) / (med(dec_ms) / 1e3),
        "images_per_sec": n_tiles / (med(vit_ms) / 1e3),
        "ttft_ms_p50": med(vit_ms) + med(pre_ms),
        "vit_ms_p50": med(vit_ms), "prefill_ms_p50": med(pre_ms), "decode_ms_per_token_p50": med(dec_ms) / (a.gen - 1),
        "vit_mfma_frac": vit_flops / (med(vit_ms) / 1e3) / 1e12 / MFMA_PEAK_TFLOPS / world,
        "prefill_mfma_frac": pre_flops / (med(pre_ms) / 1e3) / 1e12 / MFMA_PEAK_TFLOPS / world,
        "decode_hbm_frac": (14.14e9 / world + 57344.0 * S) / (med(dec_ms) / 1e3 / (a.gen - 1)) / 1e9 / HBM_PEAK_GBS if not a.tiny else None,
        "roofline": roof, "roofline_prefill": roof_pre, "roofline_vit": roof_vit,
        "device_gb": eng.device_bytes() / 1e9,
        "decode_graph": eng.decode_graph_stats() if (a.graph and world == 1) else None,
    }
    # weight-only fp8 decode (row f-2 / configs[4]), outside the timed region: same prompt, 64 greedy tokens on the e4m3 replica
    if world == 1 and not a.no_fp8:
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
                             "hbm_frac": ((14.14e9 / 2 + 57344.0 * S) / (min(t8) / 1e3) / 1e9 / HBM_PEAK_GBS) if not a.tiny else None,
                             "note": "decoder GEMV weights as OCP e4m3 + per-row fp32 scale (7.07 GB/step instead of 14.14); "
                                     "prefill, KV cache and activations stay 16-bit; NOT part of `value`"}
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
    if not a.no_cpu_baseline and world == 1:          # CPU baseline: rank 0 at N = 1 only
        res["cpu_baseline"] = cpu_baseline(cfg, S, a.gen, n_tiles)
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
    print(json.dumps(res))


if __name__ == "__main__":
    main()
