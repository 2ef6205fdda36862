import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import neko_oracle as O
import test_policy_gpu as T
from neko_amd.training.optim import NekoAdamW
from neko_amd.training.schedulers import get_linear_warmup_cosine_decay_scheduler
f = torch.load("tests/golden/g7b_trace.pt", weights_only=False)
cfg = O.OracleConfig(**f["cfg"])
def run():
    m, _ = T.make_policy(cfg, f["seed"], train=True)
    opt = NekoAdamW(m, lr=f["lr"], betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    sch = get_linear_warmup_cosine_decay_scheduler(opt, f["warmup"], f["total_steps"], base_lr=f["lr"], init_lr=f["init_lr"], min_lr=f["min_lr"])
    batches = [T.to_dev(b) for b in f["batches"]]
    norms = []; per = []
    for step in range(f["total_steps"]):
        _, loss = m.forward(inputs=batches[step % len(batches)], compute_loss=True, return_logits=False)
        loss.backward()
        if step in (0, 1):
            per.append({k: float(p.grad.float().norm()) for k, p in m.named_parameters() if p.grad is not None})
        norms.append(opt.clip_grad_norm_(1.0)); opt.step(); sch.step(); opt.zero_grad()
    return torch.stack(norms).reshape(-1).cpu().tolist(), per
ref = f["trace"]["grad_norm"]
runs = [run() for _ in range(4)]
for n, _ in runs:
    rel = [abs(a - b) / b for a, b in zip(n, ref)]
    top = sorted(range(100), key=lambda i: -rel[i])[:4]
    print("max", max(rel), "at", top, [round(rel[i], 4) for i in top], "step0/1 rel", round(rel[0], 5), round(rel[1], 5))
# which parameters differ between runs at step 0 (same weights, same batch -> only nondeterminism)
a, b = runs[0][1][0], runs[1][1][0]
d = sorted(((abs(a[k] - b[k]) / max(a[k], 1e-12), k, a[k]) for k in a), reverse=True)[:6]
print("step-0 run-to-run per-parameter grad-norm differences:", [(round(x, 5), k, round(v, 4)) for x, k, v in d])
