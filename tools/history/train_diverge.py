"""Loss curve of the cfg4 training step: python tools/train_diverge.py [bf16|fp32] [steps] [lr] [graph|eager]
Prints loss, the eight terms, beta and skin_aux every `every` steps (the bf16 step printed loss 153)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_helpers import TrainHarness, TRAIN_TERMS

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-4
mode = sys.argv[4] if len(sys.argv) > 4 else "graph"
every = int(os.environ.get("EVERY", "20"))
h = TrainHarness(precision=prec, lr=lr)
if mode == "graph":
    h.capture()
print(f"# {prec} lr={lr} {mode}")
for i in range(steps):
    h.step()
    if i % every == 0 or i == steps - 1:
        t = h.terms.tolist()
        print(f"step {h.steps_done:4d} loss {h.loss():10.4f} " + " ".join(f"{n}={v:.4g}" for n, v in zip(TRAIN_TERMS, t)) +
              f" beta={float(h.models['coarse'].beta):.5f} aux={h.models['skin_aux'].tolist()}", flush=True)
