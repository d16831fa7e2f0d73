"""C5 decode with top-k sampling on the device: the sampling head as the last phase of the persistent launch (SPN_DEC_PAIR_SAMPLE=1, default)
against the sampling head in its own launch (=0): us per note over one RenderSession.decode call -- python tools/bench_dec_sampling.py [L]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.decode import RenderSession
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import PREDICTED_DIMS, model_config, synthetic_batch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda")
torch.manual_seed(0)
model = ScorePerformer.init(model_config("c5", max_seq_len=L)); ParamArena(model, dev); model.eval()
batch = synthetic_batch(1, L, seed=7, device=dev)
with torch.no_grad():
    enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                 bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"], compute_loss=False)
truth, masked = batch["perf"][0], batch["masked_perf"][0]
ctx, sty = enc.score_embeddings[0], enc.perf_embeddings[0]
dims = list(PREDICTED_DIMS)
res = {}
for flag in ("0", "1"):
    os.environ["SPN_DEC_PAIR_SAMPLE"] = flag
    sess = RenderSession(model.perf_decoder.model, L, dims)
    sess.configure(dict(k=None, thres=0.9, temperature=1.0, seed=3))
    best = None
    for rep in range(2):
        sess.reset()
        win = truth.clone(); win[1:, dims] = 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rows = sess.decode(win, masked, ctx, sty, L - 1, batched_prefill=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res[flag] = (best / (L - 1) * 1e6, rows.clone(), sess.pair_head, sess.pair_embed, sess.pair_chains[0].max_notes if sess.pair_chains else 0)
    print(f"SPN_DEC_PAIR_SAMPLE={flag}: {res[flag][0]:.1f} us per note  head in the launch {res[flag][2]}  embed {res[flag][3]}  notes per launch {res[flag][4]}", flush=True)
print("tokens identical", bool(torch.equal(res["0"][1], res["1"][1])))
