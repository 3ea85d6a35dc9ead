#!/usr/bin/env python3
"""Gradient / logit error of the model against the reference-generated fixtures (tests/golden/model_*.npz) for each
precision setting of the nn.Linear GEMMs (sigma_amd/gemm.py): vendor fp32, split-operand MFMA with 2 or 3 bf16 pieces,
per GEMM kind (forward, input gradient, weight gradient).  Prints one JSON line per (fixture, setting):
worst element-wise gradient error relative to the tensor's max (the quantity tests/test_model_gpu.py bounds by 2e-3),
worst digest error relative to the L1 mass (bounded by 5e-3), logit error relative to the logit scale (bounded by 1e-3)."""
import json
import os
import sys

os.environ["SIGMA_GEMM"] = "split3"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sigma_amd import gemm  # noqa: E402
from tests.model_utils import build_model, digest, fill, load_model_golden  # noqa: E402

SETTINGS = [("fp32", 0, 0, 0), ("f2", 2, 0, 0), ("f2_d2", 2, 2, 0), ("f2_w2", 2, 0, 2), ("f2_d2_w2", 2, 2, 2), ("f2_d3_w2", 2, 3, 2),
            ("f2_d3_w3", 2, 3, 3), ("f3_d3_w3", 3, 3, 3), ("f3_d3_w2", 3, 3, 2)]


def run(case, fwd, dgrad, wgrad):
    gemm._FWD, gemm._DGRAD, gemm._WGRAD = fwd, dgrad, wgrad
    meta, z = load_model_golden(case)
    torch.manual_seed(0)
    model = build_model(meta["backbone"], meta["num_classes"], meta["H"], meta["W"]).cuda().eval()
    rgb, x, label = fill.make_inputs(meta["batch"], meta["H"], meta["W"], meta["num_classes"])
    with torch.no_grad():
        logits = model(rgb.cuda(), x.cuda())
    ref = torch.from_numpy(z["logits"])
    logit_err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
    loss = model(rgb.cuda(), x.cuda(), label.cuda())
    loss.backward()
    got = dict(model.named_parameters())
    worst_d, dlist = 0.0, []
    for n, r in zip(list(z["grad_names"]), z["grad_digest"]):
        d = digest(got[n].grad)
        e = max(abs(float(d[i]) - float(r[i])) for i in range(3)) / (abs(float(r[1])) + 1e-6)
        dlist.append((e, str(n)))
        worst_d = max(worst_d, e)
    dlist.sort(reverse=True)
    worst_e, worst_n = 0.0, ""
    for i, n in enumerate(list(z["grad_full_names"])):
        r = torch.from_numpy(z[f"grad_full_{i}"])
        g = got[str(n)].grad.cpu()
        e = float((g - r).abs().max()) / (float(r.abs().max()) + 1e-7)
        if e > worst_e:
            worst_e, worst_n = e, str(n)
    return dict(top_digest=[(round(e, 6), n) for e, n in dlist[:4]], logit_err=logit_err, loss_err=abs(loss.item() - float(z["loss"])), worst_digest=worst_d, worst_elem=worst_e, worst_elem_name=worst_n)


def main():
    settings = SETTINGS
    if len(sys.argv) > 1:            # e.g. f3_d2_w2 f3_d3_w3 f3_d3_w3
        def parse(tok):
            if tok == "fp32":
                return ("fp32", 0, 0, 0)
            parts = dict((p[0], int(p[1:])) for p in tok.split("_"))
            return (tok, parts.get("f", 0), parts.get("d", 0), parts.get("w", 0))
        settings = [parse(t) for t in sys.argv[1:]]
    for case in ("tiny_64x96", "tiny_72x88_b2"):
        for name, f, d, w in settings:
            rec = dict(case=case, setting=name, **run(case, f, d, w))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
