#!/usr/bin/env python3
"""Per-kernel roofline table of the BASELINE configs[1] sample from ONE rocprofv3 kernel trace (+ optional PMC passes).

    rocprofv3 --kernel-trace --output-format csv -d D -o run -- python3 bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 ...
    python3 tools/roofline_table.py D/**/run_kernel_trace.csv [--fetch FETCH_counter_collection.csv --write WRITE_counter_collection.csv]

The kernel-stats CSV groups by kernel NAME, and one name serves several call sites (the 256x256 GEMM with a plain epilogue runs the ViT qkv,
the prefill qkv and the projector).  Here every dispatch is labelled by its ROLE from its own kind and its neighbours in launch order (the
launch sequence of a layer is fixed: model.hip vit_run / prefill / decode_body), the roles are priced with the algorithmic work of
DESIGN.md section 4 at the given geometry, and `achieved / peak` is printed per role -- the table VERDICT r04 built by hand.
PMC: FETCH_SIZE / WRITE_SIZE in KiB per dispatch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: traffic =
2 x FETCH + WRITE.  The PMC passes are separate runs of the same command; their dispatches are labelled by the same rules."""
import argparse
import collections
import csv
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_roles import FAMILIES      # one table of kernel names for bench.py and this tool (tools/kernel_roles.py)

HBM_PEAK = 8000.0      # GB/s
MFMA_PEAK = 2500.0     # TFLOP/s dense bf16


def kind_of(name):
    """(family, epilogue code or None) of a kernel name (mangled or demangled)"""
    for fam in FAMILIES:
        if fam in name:
            epi = None
            if fam.startswith("gemm8"):
                m = re.search(fam + r"I[A-Za-z0-9]+?Li(\d)E", name)
                epi = int(m.group(1)) if m else None
            elif fam == "gemm_kernel":
                m = re.search(r"gemm_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d)E", name)
                epi = int(m.group(5)) if m else None
            elif fam == "gemv_rows_norm_kernel":
                m = re.search(r"gemv_rows_norm_kernelI[A-Za-z0-9_]+?Li(\d)E", name)
                epi = int(m.group(1)) if m else None
            elif fam == "attn2_kernel":
                m = re.search(r"attn2_kernelI\w+?Li(\d)ELb(\d)E", name)
                epi = int(m.group(2)) if m else None           # 1 = causal (decoder prefill), 0 = the ViT
            return fam, epi
    return None, None


def label(seq):
    """seq: list of kernel names in launch order -> list of role strings (None for kernels outside the hot path)"""
    kinds = [kind_of(n) for n in seq]
    fam = [k[0] for k in kinds]
    out = [None] * len(seq)
    ctx = None
    for i, (f, e) in enumerate(kinds):
        prev = fam[i - 1] if i else None
        nxt = fam[i + 1] if i + 1 < len(seq) else None
        if f == "attn2_kernel":
            ctx = "pre" if e == 1 else "vit"
            out[i] = "prefill attention (causal GQA)" if e == 1 else "ViT attention (MHA)"
        elif f in ("attn_decode_kernel", "attn_decode_dma_kernel", "attn_decode_multi_kernel", "attn_decode_kv8_walk_kernel"):
            ctx = "dec"
            out[i] = "decode attention (split-KV)"
        elif f in ("attn_merge_kernel", "attn_merge_mid_kernel"):
            out[i] = "decode attention merge"
        elif f == "vit_qknorm_kernel":
            out[i] = "ViT q/k norm"
        elif f == "vit_knorm_slots_kernel":
            out[i] = "ViT K norm (from the qkv slots; leaves the q sums)"
        elif f in ("gemm8_kernel", "gemm8p_kernel", "gemm_kernel"):
            nxt2 = fam[i + 2] if i + 2 < len(seq) else None
            if nxt == "vit_knorm_slots_kernel":                 # round 6, fused layer: qkv (row scale + statistics) -> K norm -> attention
                ctx = "vit"
                out[i] = "ViT qkv GEMM"
            elif ctx == "vit" and i and out[i - 1] == "ViT proj GEMM" and prev in ("gemm8_kernel", "gemm8p_kernel", "gemm_kernel"):
                out[i] = "ViT fc1 GEMM (GELU)"                    # fused layer: no norm launch between proj and fc1
            elif nxt in ("vit_qknorm_kernel", "vit_qk_sumsq_kernel") or (nxt in ("gemm8_kernel", "gemm8p_kernel", "gemm_kernel") and nxt2 in ("vit_qknorm_kernel", "vit_qk_sumsq_kernel")
                                                                       and prev == "rmsnorm_kernel"):
                out[i] = "ViT qkv GEMM"
            elif nxt == "rope_kv_kernel":
                out[i] = "prefill qkv GEMM"
            elif prev == "attn2_kernel":
                out[i] = "ViT proj GEMM" if ctx == "vit" else "prefill o_proj GEMM"
            elif prev == "rmsnorm_kernel" and nxt in ("gemm8_kernel", "gemm8p_kernel", "gemm_kernel"):
                # fc1 / gate|up follow a norm and feed the second MLP GEMM; the ViT's follows its attention block
                out[i] = "ViT fc1 GEMM (GELU)" if ctx == "vit" else "prefill gate|up GEMM (SwiGLU)"
            elif prev in ("gemm8_kernel", "gemm8p_kernel", "gemm_kernel") and out[i - 1] in ("ViT fc1 GEMM (GELU)", "prefill gate|up GEMM (SwiGLU)"):
                # a GEMM may run as TWO launches over disjoint column ranges (tile ids 12 / 13: whole rounds + a tail): the tail keeps the epilogue
                # of its GEMM (GELU = 1 for fc1), the next GEMM of the MLP has the residual epilogue (2 / 3)
                if e == 1 and out[i - 1] == "ViT fc1 GEMM (GELU)":
                    out[i] = out[i - 1]
                else:
                    out[i] = "ViT fc2 GEMM" if ctx == "vit" else "prefill down_proj GEMM"
            elif prev in ("gemm8_kernel", "gemm8p_kernel", "gemm_kernel") and out[i - 1] in ("ViT fc2 GEMM", "prefill down_proj GEMM", "ViT qkv GEMM", "ViT proj GEMM",
                                                                                         "prefill qkv GEMM", "prefill o_proj GEMM") and kinds[i - 1][1] == e and e is not None:
                out[i] = out[i - 1]                              # tail launch of the same GEMM
        elif f == "gemv_rows_norm_loop_kernel":
            out[i] = "decode gate|up GEMV (+RMSNorm)"
        elif f == "gemv_rows_longk_kernel":
            out[i] = "decode down_proj GEMV"
        elif f == "gemv_rows_norm_kernel":
            if e == 4:                                          # EPI_SWIGLU: the non-loop gate|up form (round 5 default)
                out[i] = "decode gate|up GEMV (+RMSNorm)"
            else:
                out[i] = "decode qkv GEMV (+RMSNorm)" if nxt and nxt.startswith("attn_decode") else ("lm_head GEMV (+final norm)" if nxt and nxt.startswith("argmax") else None)
        elif f == "gemv_rows_kernel" and prev in ("attn_merge_kernel", "attn_merge_mid_kernel"):
            out[i] = "decode o_proj GEMV"
        elif f == "rmsnorm_kernel":
            rope_ahead = "rope_kv_kernel" in fam[i + 1:i + 3]
            out[i] = "RMSNorm (prefill)" if (ctx == "pre" or rope_ahead) else "RMSNorm (ViT)"
        elif f == "rope_kv_kernel":
            out[i] = "prefill RoPE + KV write"
    return out


def work(tiles, text, gen):
    """role -> (unit, algorithmic work per launch).  OmChat-13B (omchat_amd/config.py omchat13b); DESIGN.md section 4."""
    C, I, Hv, ntok = 3200, 12800, 25, 1025
    H, It, qkvd, qd, V = 3584, 18944, 4608, 3584, 152064
    M = tiles * ntok
    S = tiles * 1024 + text
    L = S + gen / 2.0
    F, B = "TFLOP/s", "GB/s"
    return {
        "ViT qkv GEMM": (F, 2.0 * M * 3 * C * C), "ViT proj GEMM": (F, 2.0 * M * C * C), "ViT fc1 GEMM (GELU)": (F, 2.0 * M * I * C),
        "ViT fc2 GEMM": (F, 2.0 * M * C * I), "ViT attention (MHA)": (F, 4.0 * tiles * Hv * ntok * ntok * 128),
        "ViT q/k norm": (B, M * 2 * C * 2 * 2.0), "ViT K norm (from the qkv slots; leaves the q sums)": (B, M * C * 2 * 2.0), "RMSNorm (ViT)": (B, M * C * 2 * 2.0), "RMSNorm (prefill)": (B, S * H * 2 * 2.0),
        "prefill qkv GEMM": (F, 2.0 * S * qkvd * H), "prefill o_proj GEMM": (F, 2.0 * S * H * qd), "prefill gate|up GEMM (SwiGLU)": (F, 2.0 * S * 2 * It * H),
        "prefill down_proj GEMM": (F, 2.0 * S * H * It), "prefill attention (causal GQA)": (F, 4.0 * S * S * 128 * 28 / 2),
        "prefill RoPE + KV write": (B, S * (qkvd * 2.0 * 2 + 1024 * 2.0)),
        "decode qkv GEMV (+RMSNorm)": (B, qkvd * H * 2.0), "decode o_proj GEMV": (B, H * qd * 2.0), "decode gate|up GEMV (+RMSNorm)": (B, 2.0 * It * H * 2),
        "decode down_proj GEMV": (B, H * It * 2.0), "decode attention (split-KV)": (B, L * 2048.0), "decode attention merge": (B, None),
        "lm_head GEMV (+final norm)": (B, V * H * 2.0),
    }


def read_trace(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [r["Kernel_Name"] for r in rows], [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]


def read_pmc(path, counter):
    """per-dispatch counter values in launch order (rows of one dispatch may repeat per XCD / dimension: summed)"""
    agg, names, order = collections.OrderedDict(), {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        agg[d] = agg.get(d, 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
        order[d] = int(r.get("Start_Timestamp") or d)
    ids = sorted(agg, key=lambda d: (order[d], d))
    return [names[d] for d in ids], [agg[d] for d in ids]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--fetch"); ap.add_argument("--write")
    ap.add_argument("--tiles", type=int, default=3); ap.add_argument("--text", type=int, default=512); ap.add_argument("--gen", type=int, default=32)
    a = ap.parse_args()
    names, us = read_trace(a.trace)
    roles = label(names)
    W = work(a.tiles, a.text, a.gen)
    t = collections.defaultdict(list)
    prev_role = None
    for r, u in zip(roles, us):
        if r:
            if r == prev_role and "GEMM" in r:
                t[r][-1] += u                      # the tail launch of a two-launch GEMM: one call
            else:
                t[r].append(u)
        prev_role = r
    traffic = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
    for path, ctr, slot, mul in ((a.fetch, "FETCH_SIZE", 0, 2.0), (a.write, "WRITE_SIZE", 1, 1.0)):
        if path:
            pn, pv = read_pmc(path, ctr)
            pr = None
            for r, v in zip(label(pn), pv):
                if r:
                    traffic[r][slot] += mul * v * 1024.0
                    if not (r == pr and "GEMM" in r):
                        traffic[r][2 + slot] += 1
                pr = r
    total = sum(us)
    print(f"# {a.trace}: {len(us)} dispatches, {total / 1e3:.2f} ms of kernel time; geometry: {a.tiles} tiles + {a.text} text ids, {a.gen} decode tokens")
    print(f"{'role':52s} {'calls':>6s} {'avg us':>9s} {'% time':>7s} {'work / launch':>16s} {'achieved':>14s} {'frac':>6s} {'PMC traffic':>12s} {'x alg.':>7s}")
    for r, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
        unit, w = W.get(r, (None, None))
        avg = sum(v) / len(v)
        if w is None:
            ach = frac = wtxt = ""
        elif unit == "TFLOP/s":
            x = w / avg / 1e6
            ach, frac, wtxt = f"{x:8.1f} TF/s", f"{x / MFMA_PEAK:.3f}", f"{w / 1e9:10.2f} GF"
        else:
            x = w / avg / 1e3
            ach, frac, wtxt = f"{x:8.1f} GB/s", f"{x / HBM_PEAK:.3f}", f"{w / 1e6:10.2f} MB"
        tr = traffic.get(r)
        ttxt = rtxt = ""
        if tr and tr[2] and (tr[3] or not a.write):
            per = tr[0] / tr[2] + (tr[1] / tr[3] if tr[3] else 0.0)
            ttxt = f"{per / 1e6:9.1f} MB"
            if w and unit == "GB/s":
                rtxt = f"{per / w:.2f}"
        print(f"{r:52s} {len(v):6d} {avg:9.2f} {100 * sum(v) / total:7.2f} {wtxt:>16s} {ach:>14s} {frac:>6s} {ttxt:>12s} {rtxt:>7s}")
    other = total - sum(sum(v) for v in t.values())
    print(f"{'(unlabelled: fills, copies, argmax, front end)':52s} {'':6s} {'':9s} {100 * other / total:7.2f}")


if __name__ == "__main__":
    main()
