"""Same-box A/B of the C5 greedy decode (L notes, batch 1) under environment flags of the decode engine, interleaved:
    python tools/bench_dec_flags.py 4096 SPN_DEC_PAIR_HEAD=0 SPN_DEC_PAIR_HEAD=1 [...]
Every variant is "NAME=value[,NAME=value]" (NAME = an environment variable, or tune.<knob> for a knob of csrc/tuning.h, which keeps
its value until another variant sets it); tokens must be identical across variants."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def main():
    from scoreperformer_amd import lib
    from scoreperformer_amd.arena import ParamArena
    from scoreperformer_amd.models import ScorePerformer
    from scoreperformer_amd.modules.sampling import top_k
    from scoreperformer_amd.synthetic import model_config, synthetic_batch
    L = int(sys.argv[1])
    variants = sys.argv[2:]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = ScorePerformer.init(model_config("c5", max_seq_len=L))
    ParamArena(model, dev)
    model.eval()
    batch = synthetic_batch(1, L, seed=7, device=dev)
    with torch.no_grad():
        enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                     bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"],
                                     compute_loss=False)
    tokens = batch["masked_perf"].clone()
    tokens[:, 0] = batch["perf"][:, 0]
    dec = model.perf_decoder
    names = sorted({kv.split("=")[0] for v in variants for kv in v.split(",")})
    res, outs = {v: [] for v in variants}, {}
    for rnd in range(3):
        for v in variants:
            for n in names:
                os.environ.pop(n, None)
            for kv in v.split(","):
                k, _, val = kv.partition("=")
                if k.startswith("tune."):           # a knob of csrc/tuning.h (read at every launch / graph capture)
                    lib.set_tuning(k[5:], float(val))
                else:
                    os.environ[k] = val
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = dec.unmask_tokens(tokens, batch["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
                                    filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / (L - 1) * 1e6)
            outs[v] = out.cpu()
    first = variants[0]
    for v in variants:
        print(f"{v:40s} {min(res[v]):7.1f} us per note (runs: {', '.join(f'{x:.1f}' for x in res[v])})  tokens equal to '{first}': {torch.equal(outs[v], outs[first])}",
              flush=True)


main()
