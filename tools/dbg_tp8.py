"""Debug: is an N-rank tensor-parallel run on ONE GPU (rank processes over the IPC peer transport) repeatable and equal across ranks?
   python tools/dbg_tp8.py [ranks] [dtype] [repeats]   -- full OmChat-13B geometry, configs[1] inputs.  Prints, per phase (ViT features,
   prefill logits, decode logits of 4 steps), whether every repeat gives the same bits on every rank, and the error against repeat 0."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import socket
import numpy as np
import torch


def proc(rank, size, port, q, dtype, reps):
    import ctypes as C
    import torch.distributed as dist
    from omchat_amd import synth, tp, _lib
    from omchat_amd.config import omchat13b
    from omchat_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        cfg = omchat13b()
        peer = tp.init_peer(rank, size, cap_bytes=64 << 20)
        ok, detail = tp.peer_selftest(peer, rank, size)
        S = 3 * cfg.num_image_tokens + 512
        e = Engine(cfg, dtype=dtype, max_seq=S + 48, max_batch=1, max_tiles=3, max_prefill_rows=S, tp_rank=rank, tp_size=size, comm=None)
        e.set_peer(peer, 0, all_sizes=True)
        e.fill_synthetic(0)
        px = torch.from_numpy(synth.pixels(3, cfg.vision["image_size"], 0)).to("cuda", e.torch_dtype)
        text = synth.token_ids(512, 151643, 1).tolist()
        row = []
        for t in range(3): row += [-200, text[t]]
        ids = torch.tensor([row[:-1] + text[2:]], dtype=torch.int64)
        out = []
        for r in range(reps):
            feats = e.encode_images(px)
            embeds, lengths, _ = e.splice(ids, None, feats)
            logits, _ = e.prefill(embeds, lengths)
            dec = []
            tok = e.argmax(logits)
            for _ in range(4):
                tok, lg = e.decode_step(tok, want_logits=True)
                dec.append(lg.float().cpu().numpy().copy())
            torch.cuda.synchronize()
            out.append((feats.float().cpu().numpy().copy(), logits.float().cpu().numpy().copy(), np.stack(dec)))
            dist.barrier()
        err = C.c_int(0)
        _lib.check(_lib.lib().omchat_peer_error(peer, C.byref(err)))
        q.put((rank, ok, err.value, out))
        dist.barrier()
        e.close()
    except BaseException:      # noqa
        import traceback
        q.put((rank, False, -1, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=proc, args=(r, size, port, q, dtype, reps)) for r in range(size)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=1500) for _ in ps], key=lambda x: x[0])
    for p in ps: p.join(timeout=60)
    for r in res:
        if not isinstance(r[3], list):
            print("rank", r[0], "FAILED", r[3]); sys.exit(1)
    print(f"{size} ranks, {dtype}: selftest", [r[1] for r in res], "peer timeouts", [r[2] for r in res])
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
    names = ("vit features", "prefill logits (rank shard)", "decode logits (rank shard)")
    for ph in range(3):
        for rep in range(reps):
            across = all(np.array_equal(res[0][3][rep][0], res[k][3][rep][0]) for k in range(size)) if ph == 0 else None
            vs0 = [rel(res[k][3][rep][ph], res[k][3][0][ph]) for k in range(size)]
            print(f"  {names[ph]:30s} repeat {rep}: equal across ranks {across}   rel err vs repeat 0 per rank {['%.2e' % v for v in vs0]}")
