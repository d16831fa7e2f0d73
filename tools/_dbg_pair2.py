import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch
from scoreperformer_amd.decode import GreedyDecoder
dev = torch.device("cuda:0")
torch.manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 66
model = ScorePerformer.init(model_config("c5", max_seq_len=L, depths=(1, 1, int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
ParamArena(model, dev); model.eval()
batch = synthetic_batch(1, L, seed=7, device=dev)
with torch.no_grad():
    enc = model.forward_encoders(perf=batch["perf"], perf_mask=batch["perf_mask"], score=batch["score"], score_mask=batch["score_mask"],
                                 bars=batch["bars"], beats=batch["beats"], onsets=batch["onsets"], deadpan_mask=batch["deadpan_mask"], compute_loss=False)
tokens = batch["masked_perf"].clone(); tokens[:, 0] = batch["perf"][:, 0]
os.environ["SPN_DEC_PAIR"] = "0"
e0 = GreedyDecoder(model.perf_decoder.model, L, use_graph=False); e0.run(tokens, batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
os.environ["SPN_DEC_PAIR"] = "1"
e1 = GreedyDecoder(model.perf_decoder.model, L, use_graph=False); e1.run(tokens, batch["masked_perf"], enc.score_embeddings, enc.perf_embeddings)
torch.cuda.synchronize()
lo = lambda g: (g & 0xffffffff).to(torch.int32).view(torch.float32)
print("steps", e0.n_steps, "last position", int(e0.pos.item()) - 1)
print("qkv   max diff", float((e0.qkv - lo(e1.pair_g["gq"])).abs().max()))
p0, p1 = e0.att_part.view(-1, 66), lo(e1.pair_g["gp"]).view(-1, 66)
dp = (p0 - p1).abs()
print("parts max diff", float(dp.max()), "records differing", dp.max(dim=1).values.nonzero().flatten().tolist()[:20])
r = int(dp.max(dim=1).values.argmax())
print("record", r, "m,l", p0[r, :2].tolist(), p1[r, :2].tolist(), "first cols", p0[r, 2:6].tolist(), p1[r, 2:6].tolist())
print("g     max diff", float((e0.g - lo(e1.pair_g["gg"])).abs().max()))
print("x     max diff", float((e0.x - e1.x).abs().max()))
for i, (a, b) in enumerate(zip(e0.hid, e1.hid)):
    dif = (a - b).abs().max(dim=1).values; nz = dif.nonzero().flatten()
    print("hid", i, "first differing position", int(nz[0]) if len(nz) else None, "max", float(dif.max()))
