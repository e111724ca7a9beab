#!/usr/bin/env python3
"""Projected tensor-parallel scaling of bench.py's workloads from ONE-GPU measurements: per-rank compute time comes
from `bench.py --shard-of N` lines (rank 0's shard shapes and launch sequence with the exchanges removed), the exchanges are priced
from their message bytes.  No multi-GPU node was available to the builder: this is arithmetic on measured inputs, not a measurement.

    python tools/tp_projection.py profiles/r03_a_bench_n1.json profiles/r03_a_bench_shard2.json profiles/r03_a_bench_shard4.json profiles/r03_a_bench_shard8.json

Communication model (stated, not measured):
  * xGMI: 7 links x 153 GB/s per GPU and direction (BASELINE / MI355X guide).  A ring / direct all-reduce of B bytes moves 2 (n-1)/n B per
    GPU; with all (n-1) links of the fully connected node busy the bound is t = 2 (n-1)/n B / ((n-1) x 153 GB/s) = 2 B / (n x 153 GB/s).
    RCCL is priced at EFF = 60 % of that bound plus LAT_RCCL = 25 us per call (launch + protocol).
  * decode-sized messages go through the one-shot peer kernel (csrc/comm.hip) fused with the residual + RMSNorm launch that exists anyway:
    extra cost LAT_PEER = 4 us per exchange (flag round trip over xGMI + (n-1) remote 16-byte-granule reads), 2 per layer + 1 for the argmax.
  * sequence-parallel norms (round 6, model.hip gemm_sp; shard lines whose comm_stats carry sp_reduce_scatters > 0): an exchange is a
    reduce-scatter + an all-gather over row blocks -- the two halves of the all-reduce, the same bytes, TWO calls (2 x LAT_RCCL); what it
    buys is in the per-rank compute of the shard lines (the norms run on rows / n).  The final gather of the residual stream (one per tower
    chunk / per prefill) is priced as half an all-reduce.
  * prefill / ViT all-reduces run on the communication stream under the next row chunk's GEMM (model.hip: gemm_allreduce): per
    projection the exposed time is max(sum of chunk all-reduces - GEMM time of the later chunks, one chunk's all-reduce); projections
    too small to be chunked (a chunk must hold >= 256 tiles: the 3-tile sample, the single-sequence prefill) expose all of it.
"""
import json
import sys

LINK = 153e9
EFF = 0.60
LAT_RCCL = 25e-6
LAT_PEER = 4e-6
F_ROW = 0.41          # share of a phase's per-rank compute spent in the two row-parallel projections (proj + fc2 / o_proj + down_proj)
# sequence-parallel form: the consumer of a gathered activation (the next column-parallel GEMM: qkv / fc1, qkv / gate|up) is issued per row chunk
# behind that chunk's all-gather when a chunk holds two whole rounds of its tiles (model.hip gemm_after_sp), so the exchange of chunk i + 1 also
# hides under the consumer's chunk i.  Share of a phase's compute in each consumer (TP = 1 kernel stats, profiles/r05_ad_roofline_table_configs1.txt):
F_CONS = {"vit": {"qkv": 0.19, "fc1": 0.29}, "pre": {"qkv": 0.06, "gu": 0.51}}
N_COLS = {"vit": {"qkv": 9600, "fc1": 12800}, "pre": {"qkv": 4608, "gu": 37888}}


def allreduce_s(nbytes, n, calls=1):
    return calls * LAT_RCCL + 2.0 * nbytes / (n * LINK) / EFF


def main():
    files = sys.argv[1:]
    base = json.load(open(files[0]))
    shards = {}
    for f in files[1:]:
        d = json.load(open(f))
        shards[d["shard_of"]] = d
    print("workload      N   ViT ms  prefill ms  decode ms/step | comm: ViT  prefill  decode/step |  step ms   speed-up   (DP tower: step ms, speed-up)")
    for wl, key, b, tiles_chunk, n_chunks in (("configs1", None, 1, 3, 1), ("configs2", "configs2", 32, 24, 4)):
        d1 = base if key is None else base[key]
        dec1 = d1.get("decode_ms_per_token_p50", d1.get("decode_ms_per_step_p50"))
        gen = 256
        t1 = d1["vit_ms_p50"] + d1["prefill_ms_p50"] + (gen - 1) * dec1
        print(f"{wl:10s}    1  {d1['vit_ms_p50']:7.1f}  {d1['prefill_ms_p50']:9.1f}  {dec1:12.3f} |      -        -         -       | {t1:8.1f}      1.00")
        for n in sorted(shards):
            s = shards[n] if key is None else shards[n][key]
            dec = s.get("decode_ms_per_token_p50", s.get("decode_ms_per_step_p50"))
            sp = bool((shards[n].get("comm_stats") or {}).get("sp_reduce_scatters"))
            calls = 2 if sp else 1                                   # reduce-scatter + all-gather, or one all-reduce
            S = 3584
            # ViT: 45 layers x 2 all-reduces of [tiles x 1025, 3200] 16-bit per tower chunk (+ the [rows, 2] fp32 q/k-norm sums)
            m_vit = tiles_chunk * 1025 * 3200 * 2
            # prefill: 28 layers x 2 all-reduces of [b x S, 3584] 16-bit
            m_pre = b * S * 3584 * 2
            # overlap (model.hip: gemm_allreduce): a projection whose output holds >= 256 tiles per chunk is cut into 4 row chunks and the
            # all-reduce of chunk i runs under the GEMM of chunk i + 1, after which the launch stream waits for the last all-reduce:
            # exposed = max(sum of the chunk all-reduces - 3/4 of that projection's GEMM time, one chunk's all-reduce).  The GEMM time of
            # the two row-parallel projections is F_ROW of the phase's compute (kernel stats, profiles/r03_a_kernel_stats_shard8_*).
            def exposed(n_ar, msg_bytes, phase_ms, chunked, phase="vit", rows_chunk=0):
                one = allreduce_s(msg_bytes, n, calls)
                if not chunked:
                    return n_ar * one
                g = F_ROW * phase_ms / 1e3 / n_ar                      # GEMM seconds of one projection
                total = 4 * calls * LAT_RCCL + (one - calls * LAT_RCCL)  # four chunk exchanges, the same bytes
                if not sp:
                    return n_ar * max(total - 0.75 * g, total / 4)
                # sequence-parallel: half of the exchanges feed each of the two consumers; a consumer that is issued per chunk hides 3/4 of its
                # own time too, and then only the FIRST chunk's exchange (behind 3/4 of the producer) is the floor
                out = 0.0
                for cons, share in F_CONS[phase].items():
                    tiles = -(-rows_chunk // 256) * -(-(N_COLS[phase][cons] // n) // 256)
                    gc = share * phase_ms / 1e3 / (n_ar / 2) if tiles >= 512 else 0.0
                    out += (n_ar / 2) * max(total - 0.75 * (g + gc), (total / 4 - 0.75 * g) if gc else total / 4, 0.0)
                return out
            chunked = b > 1
            exp_vit = exposed(45 * n_chunks * 2, m_vit, s["vit_ms_p50"], chunked, "vit", tiles_chunk * 1025 // 4) + 45 * n_chunks * allreduce_s(tiles_chunk * 1025 * 8, n)
            exp_pre = exposed(28 * 2, m_pre, s["prefill_ms_p50"], chunked, "pre", b * S // 4)
            if sp:      # the row-sharded residual stream made whole once per tower chunk / per prefill: an all-gather = half an all-reduce
                exp_vit += n_chunks * (LAT_RCCL + 0.5 * (allreduce_s(m_vit, n) - LAT_RCCL))
                exp_pre += LAT_RCCL + 0.5 * (allreduce_s(m_pre, n) - LAT_RCCL)
            dec_comm = (28 * 2 + 1) * LAT_PEER
            vit = s["vit_ms_p50"] + exp_vit * 1e3
            pre = s["prefill_ms_p50"] + exp_pre * 1e3
            step = vit + pre + (gen - 1) * (dec + dec_comm * 1e3)
            # data-parallel tower: tiles / n per GPU at the TP = 1 per-tile rate + one gather of the features
            tiles = tiles_chunk * n_chunks
            vit_dp = d1["vit_ms_p50"] * max(1, -(-tiles // n)) / tiles + allreduce_s(tiles * 1024 * 3584 * 2, n) * 1e3
            step_dp = vit_dp + pre + (gen - 1) * (dec + dec_comm * 1e3)
            print(f"{wl:10s}  {n:3d}{'*' if sp else ' '} {s['vit_ms_p50']:7.1f}  {s['prefill_ms_p50']:9.1f}  {dec:12.3f} | {exp_vit*1e3:7.1f}  {exp_pre*1e3:7.1f}  {dec_comm*1e3:9.3f}     | {step:8.1f}  {t1/step:8.2f}     ({step_dp:8.1f}, {t1/step_dp:5.2f})")


    if any((d.get("comm_stats") or {}).get("sp_reduce_scatters") for d in shards.values()):
        print("(* = sequence-parallel norms: every exchange priced as reduce-scatter + all-gather)")


if __name__ == "__main__":
    main()
