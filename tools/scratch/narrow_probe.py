import sys, torch
sys.path.insert(0, ".")
from oracle import ref_cpu
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.modules.sampling import top_k
from scoreperformer_amd.synthetic import PREDICTED_DIMS, model_config, synthetic_batch
dev = torch.device("cuda:0")
vocab = {"Bar": 40, "Position": 36, "Pitch": 28, "Velocity": 36, "Duration": 37, "Tempo": 29, "TimeSig": 10,
         "PositionShift": 21, "NotesInOnset": 16, "PositionInOnset": 16, "RelOnsetDev": 45, "RelPerfDuration": 25}
for dh in (64, 32):
    from oracle.weights import filled_state_dict
    def make():
        cfg = model_config("tiny", num_tokens=vocab, one_kv_head=True)
        cfg["perf_decoder"]["transformer"]["attention"]["dim_head"] = dh
        return cfg
    torch.manual_seed(4)
    cfg = make()
    model = ScorePerformer.init(make())
    model.load_state_dict(filled_state_dict(model, seed=3))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ParamArena(model, dev)
    model.eval()
    model.perf_decoder.use_decode_engine = False
    L = 24
    batch = synthetic_batch(1, L, seed=3, num_tokens=vocab)
    gb = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():
        enc = model.forward_encoders(perf=gb["perf"], perf_mask=gb["perf_mask"], score=gb["score"], score_mask=gb["score_mask"],
                                     bars=gb["bars"], beats=gb["beats"], onsets=gb["onsets"], deadpan_mask=gb["deadpan_mask"], compute_loss=False)
        tokens = gb["masked_perf"].clone(); tokens[:, 0] = gb["perf"][:, 0]
        out = model.perf_decoder.unmask_tokens(tokens, gb["masked_perf"], context=enc.score_embeddings, style_embeddings=enc.perf_embeddings,
                                               filter_logits_fn=top_k, filter_kwargs={"k": 1}, disable_tqdm=True)
        # product, cache-free teacher-forced on its own tokens
        tf = model.perf_decoder(out, seq_masked=gb["masked_perf"], mask=gb["perf_mask"], context=enc.score_embeddings,
                                style_embeddings=enc.perf_embeddings)
    got = out.cpu()
    keys = list(tf.logits.keys())
    with torch.no_grad():
        _, ologits = ref_cpu.tuple_transformer(sd, "perf_decoder.model.", cfg["perf_decoder"], [got[:, :-1], batch["masked_perf"][:, 1:]],
                                               causal=True, mask=torch.ones(1, L - 1, dtype=torch.bool),
                                               context=enc.score_embeddings.float().cpu()[:, 1:], style=enc.perf_embeddings.float().cpu()[:, 1:], with_logits=True)
    for d in PREDICTED_DIMS:
        lg = tf.logits[keys[d]][0].float().cpu().clone(); lg[:, :2] = -float("inf")
        ol = ologits[keys[d]][0].clone(); ol[:, :2] = -float("inf")
        print(f"dh={dh} dim {d}: cached-vs-product-tf wrong {int((lg.argmax(-1) != got[0, 1:, d]).sum())}, cached-vs-oracle wrong {int((ol.argmax(-1) != got[0, 1:, d]).sum())}, "
              f"product-tf-vs-oracle logits maxdiff {float((tf.logits[keys[d]][0].float().cpu() - ologits[keys[d]][0]).abs().max()):.4f} (scale {float(ologits[keys[d]][0].abs().max()):.2f})")
