"""C5: greedy performance render, seq = 4096, batch 1 (hipGraph-replayed decode engine) vs the module path with torch.cat caches."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.modules.sampling import top_k
from scoreperformer_amd.synthetic import model_config, synthetic_batch

def main():
    L = int(os.environ.get("L", 4096))
    dev = torch.device("cuda")
    torch.manual_seed(int(os.environ.get("SEED", 0)))
    model = ScorePerformer.init(model_config("c5", max_seq_len=L)); ParamArena(model, dev); model.eval()
    batch = synthetic_batch(1, L, seed=7, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    tokens = batch["masked_perf"].clone(); tokens[:, 0] = batch["perf"][:, 0]
    dec = model.perf_decoder
    res = {}
    import scoreperformer_amd.decode as dmod
    orig_init = dmod.GreedyDecoder.__init__
    for name, use in (("engine_fused_hipgraph_fp32", True), ("engine_unfused_hipgraph_fp32", "unfused"), ("module_path_bf16_torchcat", False)):
        dmod.GreedyDecoder.__init__ = (lambda self, dec_, L_, **kw: orig_init(self, dec_, L_, fused=False, **kw)) if use == "unfused" else orig_init
        tok_ref = None
        if not use and L > 1024 and not os.environ.get("FULL"):
            Lm = 512
            t_in, m_in, ctx, sty = tokens[:, :Lm], batch["masked_perf"][:, :Lm], enc.score_embeddings[:, :Lm], enc.perf_embeddings[:, :Lm]
        else:
            Lm = L
            t_in, m_in, ctx, sty = tokens, batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings
        dec.use_decode_engine = bool(use)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = dec.unmask_tokens(t_in, m_in, context=ctx, style_embeddings=sty, filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[name] = {"notes": Lm - 1, "seconds": dt, "notes_per_s": (Lm - 1) / dt, "us_per_note": dt / (Lm - 1) * 1e6,
                     "masks_left": int((out == 1).sum())}
        res[name]["tokens_checksum"] = int(out[:, :Lm].sum())
        if name.startswith("engine"):
            res.setdefault("_engine_tokens", []).append(out.clone())
    eng = res.pop("_engine_tokens", [])
    if len(eng) == 2:
        res["fused_equals_unfused"] = bool(torch.equal(eng[0], eng[1]))
        if not res["fused_equals_unfused"]:
            diff = (eng[0] != eng[1]).any(-1)[0].nonzero().flatten()
            res["first_diff_position"] = int(diff[0]); res["positions_differing"] = int(diff.numel())
    print(json.dumps({"decode_c5": res, "seq": L}))
main()
