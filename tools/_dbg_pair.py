"""where does the pair launch differ from the five launches?  c5 decoder, 3 positions, compare stage buffers of the LAST step's FIRST divergence"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch
from scoreperformer_amd.decode import GreedyDecoder
from scoreperformer_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
model = ScorePerformer.init(model_config("c5", max_seq_len=L, **{}))
ParamArena(model, dev); model.eval()
batch = synthetic_batch(1, L, seed=7, device=dev)
with torch.no_grad():
    enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                 bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"], compute_loss=False)
tokens = batch["masked_perf"].clone(); tokens[:, 0] = batch["perf"][:, 0]
os.environ["SPN_DEC_PAIR"] = "0"
e0 = GreedyDecoder(model.perf_decoder.model, L, use_graph=(len(sys.argv) > 2)); e0.run(tokens, batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
os.environ["SPN_DEC_PAIR"] = "1"
e1 = GreedyDecoder(model.perf_decoder.model, L, use_graph=(len(sys.argv) > 2)); e1.run(tokens, batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
torch.cuda.synchronize()
for i, (a, b) in enumerate(zip(e0.hid, e1.hid)):
    dif = (a - b).abs().max(dim=1).values
    nz = dif.nonzero().flatten()
    print("hid", i, "first differing position", int(nz[0]) if len(nz) else None, "max", float(dif.max()))
for i, (a, b) in enumerate(zip(e0.kc, e1.kc)):
    dif = (a - b).abs().max(dim=1).values; nz = dif.nonzero().flatten()
    print("kc", i, "first differing position", int(nz[0]) if len(nz) else None, "max", float(dif.max()))
