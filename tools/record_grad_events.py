"""Records, on the GPU, the order in which the REAL model announces (`pending`, forward) and delivers (`ready` / `autograd`, backward)
its gradient contributions, parameter by parameter -- tied embedding tables (4 users), fused q|k|v groups, the tied LM-head projection
included -- together with the arena layout.  tests/test_dp_cpu.py replays the recording through GradSync on two gloo ranks, so the CPU
test exercises the real protocol, not a synthetic order.  Writes tests/golden/grad_events.json (data only)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scoreperformer_amd.arena import ParamArena
from scoreperformer_amd.models import ScorePerformer
from scoreperformer_amd.parallel import GradSync
from scoreperformer_amd.synthetic import model_config, synthetic_batch

dev = torch.device("cuda:0")
out = {}
for name, kw in (("tiny", {}), ("tiny_xattn_mha", dict(context_emb_mode="attention", style_emb_mode="cat", one_kv_head=False, alibi_learned=False))):
    torch.manual_seed(3)
    model = ScorePerformer.init(model_config("tiny", **kw))
    arena = ParamArena(model, dev)
    model.train()
    sync = GradSync(arena, None, bucket_mb=0.25, dry_run=True)
    batch = synthetic_batch(2, 64, seed=8, ragged=True, device=dev)
    steps = []
    for _ in range(2):
        sync.begin_step()
        arena.zero_grad()
        model(**batch).loss.backward()
        steps.append([[k, int(i)] for k, i in sync.events if k != "launch"])
        sync.finish()
    torch.cuda.synchronize()
    assert steps[0] == steps[1], "the event order must not depend on the step"
    kinds = {k for k, _ in steps[0]}
    out[name] = {"names": arena.names, "sizes": [p.numel() for p in arena.param_list], "offsets": arena.offsets, "total": arena.total,
                 "events": steps[0]}
    per = {}
    for k, i in steps[0]:
        if k == "pending":
            per[i] = per.get(i, 0) + 1
    print(name, len(arena.names), "parameters,", len(steps[0]), "events, kinds", sorted(kinds), "max contributions per parameter", max(per.values()))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grad_events.json")
if os.environ.get("SPN_OUT"):
    path = os.environ["SPN_OUT"]
with open(path, "w") as fh:
    json.dump(out, fh, separators=(",", ":"))
print("wrote", path, os.path.getsize(path), "bytes")
