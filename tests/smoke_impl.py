"""smoke(): one tiny NeSVoR training step on cuda:0 through the HIP ops, checked against the CPU oracle."""
import numpy as np
import torch


def run_smoke(device):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import small_args
    from nesvor_amd.fused import FusedTrainer
    from nesvor_amd.models import NeSVoR
    from nesvor_amd.transform import RigidTransform
    from oracle import hashgrid as hg
    from oracle import nesvor_model as nm

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = np.load(os.path.join(root, "tests", "golden", "reference_golden.npz"))
    args = small_args(device=device)
    sd = {str(k): torch.tensor(golden[f"fw_sd::{k}"]) for k in golden["fw_state_keys"]}
    model = NeSVoR(RigidTransform(sd["axisangle_init"].to(device)), torch.tensor(golden["ds_resolution"]).to(device),
                   float(golden["ds_mean"]), sd["inr.bounding_box"].to(device), args)
    model.load_state_dict(sd)
    d = lambda k: torch.tensor(golden[f"fw_{k}"]).to(device)
    trainer = FusedTrainer(model, args)
    losses = model.forward_with_noise(d("xyz"), d("v"), d("idx"), d("noise"))
    # oracle on the same inputs
    P = {k: v.clone() for k, v in sd.items()}
    bb, ax0 = P.pop("inr.bounding_box"), P.pop("axisangle_init")
    base, L = nm.grid_config(bb, args)
    levels = hg.make_levels(L, args.log2_hashmap_size, base, args.level_scale)
    ref = nm.nesvor_forward(P, levels, args, bb, torch.tensor(golden["fw_psf_sigma"]), ax0, float(golden["fw_delta"]),
                            torch.tensor(golden["fw_xyz"]), torch.tensor(golden["fw_v"]), torch.tensor(golden["fw_idx"]),
                            torch.tensor(golden["fw_noise"]))
    for k in ref:
        a, b = float(losses[k].detach()), float(ref[k])
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, (k, a, b)
    # the autograd-free evaluation of the same iteration (what train() runs): same losses, same gradients
    assert trainer.direct is not None
    l2 = trainer.direct.run(d("xyz"), d("v"), d("idx"), d("noise"))
    for k in ref:
        a, b = float(l2[k]), float(ref[k])
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, ("direct", k, a, b)
    g_direct = trainer.flat.grad.clone()
    trainer.flat.grad.zero_()
    from nesvor_amd.train import loss_weights

    w = loss_weights(args)
    sum(w[k] * losses[k] for k in losses if k in w and w[k]).backward()
    scale = float(trainer.flat.grad.abs().max())
    assert float((trainer.flat.grad - g_direct).abs().max()) <= 1e-4 * scale
    trainer.optimizer_step()
    # and the product's default issue path: the whole iteration + AdamW behind one C call (csrc/step.hip), PSF noise drawn in
    # the kernels - same data one AdamW step later and another noise stream: the data term must stay within a factor of two
    assert trainer.direct.native_ready()
    l3 = trainer.step(d("xyz"), d("v"), d("idx"))
    torch.cuda.synchronize()
    assert 0.5 * float(ref["MSE"]) <= float(l3["MSE"]) <= 2.0 * float(ref["MSE"]), (float(l3["MSE"]), float(ref["MSE"]))
    assert all(torch.isfinite(p).all() for p in model.parameters())
    print("smoke ok:", {k: float(v.detach()) for k, v in losses.items()})
