"""Golden vector for the loss of one training iteration: the reference's OWN statements, ddp-train.py:404-430
(`losses[...] = ...` for irgbl1 / vertl1 / primvolsum / kldiv and the weighted `loss = sum(...)`), executed here in
float64 on seeded stand-ins of `output` / `cudadata`, with the loss weights of the reference's configs/config.yaml:17-21
and its own `mean_ell_1` (losses.py:12-14) and `kl_loss_stable` (models/bottlenecks/vae.py:17-19).

Output: tests/golden/trainstep_loss.npz = the inputs, the loss weights, every term and the total.
tests/test_trainloop.py holds `Trainer.losses` / `Trainer.total_loss` to it, so the formula the GPU training-step parity
test replays in float64 is pinned by the reference and not by this repository.

Run in the build container only (needs /root/reference):  python tests/golden/gen_trainstep.py
Nothing of the reference's text is stored: the statements are read from the mounted file at run time; their SHA-256 is
asserted so that a change of the slice boundaries (or of the upstream file) is noticed instead of executed.
"""
import hashlib
import os
import sys

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
EXPECTED_SHA256 = "7f700b664d53452fc3776bae795c98599ff025e101a37ab270e166a2dccbbb29"  # of the sliced, dedented statements


def loss_statements():
    lines = open(os.path.join(REF, "ddp-train.py")).read().splitlines()
    first = next(i for i, l in enumerate(lines) if l.strip().startswith("losses: Dict[str, torch.Tensor] = {}"))
    start = next(i for i in range(first, len(lines)) if lines[i].strip().startswith("loss = sum("))
    depth, last = 0, None
    for i in range(start, len(lines)):
        depth += lines[i].count("(") - lines[i].count(")")
        if depth == 0:
            last = i
            break
    block = lines[first:last + 1]
    indent = len(block[0]) - len(block[0].lstrip())
    return "\n".join(l[indent:] for l in block) + "\n", (first + 1, last + 1)


def main():
    text, (l0, l1) = loss_statements()
    sha = hashlib.sha256(text.encode()).hexdigest()
    print("ddp-train.py:%d-%d sha256 %s" % (l0, l1, sha))
    if "--print-hash" in sys.argv:
        return
    assert sha == EXPECTED_SHA256, "the reference's loss statements are not the reviewed ones: re-read them, then update the hash"
    sys.path.insert(0, REF)
    from losses import mean_ell_1                      # losses.py:12-14
    from models.bottlenecks.vae import kl_loss_stable  # vae.py:17-19
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "config.yaml")))
    loss_weights = dict(cfg["train"]["losses"])

    g = torch.Generator().manual_seed(404)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    B, H, W, K, V, C = 3, 10, 12, 32, 32, 16
    vertmean, vertstd = 40.0 * rn(V, 3), torch.tensor(7.5, dtype=torch.float64)
    output = {"irgbrec": 100.0 + 40.0 * rn(B, 3, H, W), "verts": vertmean + 9.0 * rn(B, V, 3),
              "primscale": 30.0 * torch.exp(0.2 * rn(B, K, 3)), "expr_mu": 0.3 * rn(B, C), "expr_logstd": 0.2 * rn(B, C)}
    cudadata = {"image": 100.0 + 50.0 * rn(B, 3, H, W), "verts": rn(B, V, 3)}
    ns = {"torch": torch, "Dict": dict, "output": output, "cudadata": cudadata, "loss_weights": loss_weights,
          "vertstd": vertstd, "vertmean": vertmean, "mean_ell_1": mean_ell_1, "kl_loss_stable": kl_loss_stable,
          "__builtins__": {"sum": sum, "isinstance": isinstance, "tuple": tuple, "str": str,
                           "ValueError": ValueError}}
    exec(compile(text, "ddp-train.py:%d-%d" % (l0, l1), "exec"), ns)
    out = {"out/" + k: v.numpy() for k, v in output.items()}
    out.update({"data/" + k: v.numpy() for k, v in cudadata.items()})
    out.update(vertmean=vertmean.numpy(), vertstd=vertstd.numpy(), loss=ns["loss"].numpy(),
               weight_names=np.array(sorted(loss_weights)), weight_values=np.array([loss_weights[k] for k in sorted(loss_weights)]))
    for k, v in ns["losses"].items():
        out["term/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "trainstep_loss.npz"), **out)
    print("loss", float(ns["loss"]), {k: tuple(v.shape) for k, v in ns["losses"].items()}, loss_weights)


if __name__ == "__main__":
    main()
