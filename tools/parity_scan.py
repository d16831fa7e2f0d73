"""Loss parity HIP vs CPU oracle as a function of sequence length (tiny / c3 presets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu
from oracle.weights import filled_state_dict
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.synthetic import model_config, synthetic_batch

preset = os.environ.get("PRESET", "tiny")
dev = torch.device("cuda")
torch.set_num_threads(32)
for n in [int(x) for x in os.environ.get("NS", "128,256,512,1024,2048").split(",")]:
    cfg = model_config(preset, max_seq_len=max(n, 256))
    model = ScorePerformer.init(model_config(preset, max_seq_len=max(n, 256)))
    if os.environ.get("FILL", "1") == "1":
        sd = filled_state_dict(model, seed=1); model.load_state_dict(sd)
    else:
        sd = {k: v.clone() for k, v in model.state_dict().items()}
    ParamArena(model, dev); model.eval()
    batch = synthetic_batch(int(os.environ.get("B", 1)), n, seed=5)
    z = [torch.randn(256, d, generator=torch.Generator().manual_seed(i)) for i, d in enumerate(cfg["perf_encoder"]["latent_dim"])]
    model.perf_encoder._z_override = [t.to(dev) for t in z]
    with torch.no_grad():
        g = model(**{k: v.to(dev) for k, v in batch.items()})
        r = ref_cpu.score_performer_forward(sd, cfg, batch, z, training=True)
    print(f"n={n}: gpu {float(g.loss):.5f} cpu {float(r['loss']):.5f} diff {abs(float(g.loss)-float(r['loss'])):.5f}  " +
          " ".join(f"{k}:{float(g.losses[k])-float(r['losses'][k]):+.4f}" for k in r['losses'] if k in g.losses))
    hs_g = g.perf_decoder.hidden_state.float().cpu(); hs_r = r["hidden_state"]
    se_g = g.score_encoder.hidden_state.float().cpu(); se_r = r["score_embeddings"]
    pe_g = g.perf_encoder.embeddings.float().cpu(); pe_r = r["perf_embeddings"]
    print("   rel err: score_enc %.4f  perf_emb %.4f  dec_hidden %.4f" % (
        ((se_g - se_r).abs().max() / se_r.abs().max()).item(), ((pe_g - pe_r).abs().max() / pe_r.abs().max().clamp_min(1e-9)).item(),
        ((hs_g - hs_r).abs().max() / hs_r.abs().max()).item()))
