"""C5 greedy render (seq 4096, batch 1) through the hipGraph decode engine, once -- the workload of the decode kernel profile:
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 tools/prof_decode.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.modules.sampling import top_k
from scoreperformer_amd.synthetic import model_config, synthetic_batch

L = int(os.environ.get("L", 4096))
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
for rep in range(int(os.environ.get("REPS", 1))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.perf_decoder.unmask_tokens(tokens, batch["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
                                           filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"render {L - 1} notes: {dt:.3f} s = {dt / (L - 1) * 1e6:.1f} us/note, masks left {int((out == 1).sum())}")
