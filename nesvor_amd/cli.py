"""Command-line front end with the flag names and defaults of ``nesvor`` (nesvor/cli/main.py:27-326,
nesvor/cli/commands.py:64-146) for the commands that sit on the built path:

    python -m nesvor_amd.cli reconstruct   --input-stacks a.nii.gz b.nii.gz [--stack-masks ...] [--thicknesses ...]
                                           | --input-slices DIR   --output-volume v.nii.gz [--output-model m.pt]
                                           [--output-slices DIR] [--simulated-slices DIR]  [training flags]
    python -m nesvor_amd.cli register      --input-stacks a.nii.gz b.nii.gz --output-slices DIR [--registration stack|none]
    python -m nesvor_amd.cli sample-volume --input-model m.pt --output-volume v.nii.gz [--output-resolution 0.8] ...
    python -m nesvor_amd.cli sample-slices --input-model m.pt --input-slices DIR --simulated-slices DIR

Differences from the reference, all because the SVoRT transformer (pretrained weights, torchvision) is out of scope
here: ``--registration`` accepts the reference's choices, of which ``none`` (the default; the reference defaults to
``svort``) and ``stack`` (stack-to-stack rigid registration, nesvor_amd/registration.py) are implemented; there is
the ``register`` command offers the same two.  Precision follows the reference's switch
(``--single-precision`` = the fp32 model with biased Linear layers; the default is the reference's half-precision
structure - bias-free networks - which the HIP path evaluates with bf16 matrix operands and fp32 accumulation).
"""
import argparse
import logging
import os
import sys
import time
from argparse import Namespace
from typing import Dict, List

import torch


def _training_flags(p: argparse.ArgumentParser) -> None:
    g = p.add_argument_group("model architecture")
    g.add_argument("--n-features-per-level", default=2, type=int)
    g.add_argument("--log2-hashmap-size", default=19, type=int)
    g.add_argument("--level-scale", default=1.3819, type=float)
    g.add_argument("--coarsest-resolution", default=16.0, type=float)
    g.add_argument("--finest-resolution", default=0.5, type=float)
    g.add_argument("--n-levels-bias", default=0, type=int)
    g.add_argument("--depth", default=1, type=int)
    g.add_argument("--width", default=64, type=int)
    g.add_argument("--n-features-z", default=15, type=int)
    g.add_argument("--n-features-slice", default=16, type=int)
    g.add_argument("--no-transformation-optimization", action="store_true")
    g.add_argument("--no-slice-scale", action="store_true")
    g.add_argument("--no-pixel-variance", action="store_true")
    g.add_argument("--no-slice-variance", action="store_true")
    g.add_argument("--single-precision", action="store_true")
    g.add_argument("--fp16-loss-scaling", action="store_true",
                   help="(not a flag of the reference: it is the reference's DEFAULT behaviour) without --single-precision: fp16 MLP "
                        "operands and torch.cuda.amp.GradScaler semantics (init_scale 1, growth 2 every 2000 finite steps, backoff 0.5, "
                        "steps with non-finite gradients skipped) instead of this package's bf16 operands without loss scaling")
    g.add_argument("--mlp-fp16", action="store_true",
                   help="(not in the reference) with --single-precision: power-of-two-scaled fp16 MLP matrix operands (one MFMA per "
                        "product, no loss scaler needed), fp32 accumulation / weights")
    g.add_argument("--mlp-bf16", action="store_true",
                   help="(not in the reference) with --single-precision: bf16 MLP matrix operands, fp32 accumulation / weights")
    g.add_argument("--mlp-fp32-mfma", action="store_true",
                   help="(not in the reference) with --single-precision: evaluate the MLP products with fp32 MFMAs instead of the "
                        "default split-fp16 evaluation of the same fp32 products (same accuracy, slower)")
    g = p.add_argument_group("loss function")
    g.add_argument("--weight-transformation", default=0.1, type=float)
    g.add_argument("--weight-bias", default=100.0, type=float)
    g.add_argument("--image-regularization", default="edge", type=str, choices=["TV", "edge", "L2"])
    g.add_argument("--weight-image", default=2.0, type=float)
    g.add_argument("--delta", default=0.2, type=float)
    g = p.add_argument_group("training")
    g.add_argument("--learning-rate", default=5e-3, type=float)
    g.add_argument("--gamma", default=0.33, type=float)
    g.add_argument("--milestones", nargs="+", type=float, default=[0.5, 0.75, 0.9])
    g.add_argument("--n-iter", default=6000, type=int)
    g.add_argument("--batch-size", default=4096, type=int)
    g.add_argument("--n-samples", default=256, type=int)


def _output_volume_flags(g) -> None:
    g.add_argument("--output-resolution", default=0.8, type=float)
    g.add_argument("--output-intensity-mean", default=700.0, type=float)
    g.add_argument("--inference-batch-size", type=int)
    g.add_argument("--n-inference-samples", type=int)
    g.add_argument("--no-output-psf", action="store_true")


def _common_flags(p: argparse.ArgumentParser) -> None:
    g = p.add_argument_group("common")
    g.add_argument("--device", default=0, type=int, help="HIP device index")
    g.add_argument("--verbose", type=int, default=1, choices=[0, 1, 2])
    g.add_argument("--output-log", type=str)
    g.add_argument("--seed", type=int, default=None)
    g.add_argument("--debug", action="store_true")


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="nesvor_amd.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = parser.add_subparsers(dest="command", required=True)

    p = sub.add_parser("reconstruct")
    g = p.add_argument_group("input")
    g.add_argument("--input-stacks", nargs="+", type=str)
    g.add_argument("--thicknesses", nargs="+", type=float)
    g.add_argument("--stack-masks", nargs="+", type=str)
    g.add_argument("--input-slices", type=str)
    g = p.add_argument_group("output")
    g.add_argument("--output-volume", type=str)
    _output_volume_flags(g)
    g.add_argument("--output-slices", type=str)
    g.add_argument("--simulated-slices", type=str)
    g.add_argument("--output-model", type=str)
    g.add_argument("--mask-threshold", type=float, default=1.0)
    g = p.add_argument_group("registration")
    g.add_argument("--registration", default="none", type=str, choices=["svort", "svort-stack", "stack", "none"])
    g.add_argument("--svort-version", default="v1", type=str, choices=["v1", "v2"])
    _training_flags(p)
    _common_flags(p)

    p = sub.add_parser("register")
    g = p.add_argument_group("input")
    g.add_argument("--input-stacks", nargs="+", type=str, required=True)
    g.add_argument("--thicknesses", nargs="+", type=float)
    g.add_argument("--stack-masks", nargs="+", type=str)
    g = p.add_argument_group("output")
    g.add_argument("--output-slices", type=str, required=True)
    g = p.add_argument_group("registration")
    g.add_argument("--registration", default="stack", type=str, choices=["svort", "svort-stack", "stack", "none"])
    g.add_argument("--svort-version", default="v1", type=str, choices=["v1", "v2"])
    _common_flags(p)

    p = sub.add_parser("sample-volume")
    p.add_argument("--input-model", type=str, required=True)
    g = p.add_argument_group("output")
    g.add_argument("--output-volume", type=str, required=True)
    _output_volume_flags(g)
    g.add_argument("--mask-threshold", type=float, default=1.0)
    _common_flags(p)

    p = sub.add_parser("sample-slices")
    p.add_argument("--input-model", type=str, required=True)
    p.add_argument("--input-slices", type=str, required=True)
    g = p.add_argument_group("output")
    g.add_argument("--simulated-slices", type=str, required=True)
    g.add_argument("--mask-threshold", type=float, default=1.0)
    _common_flags(p)
    return parser


def merge_args(args_old: Namespace, args_new: Namespace) -> Namespace:
    """Stored (checkpoint) arguments overridden by the command line (utils/misc.py:22-26)."""
    d = dict(vars(args_old))
    d.update(vars(args_new))
    return Namespace(**d)


def stacks_to_slices(stacks) -> List:
    """``--registration none``: every non-empty slice of every stack at its nominal pose, each stack normalised by
    the 0.99 quantile of its masked intensities (svort/inference.py:553-560)."""
    slices = []
    for stack in stacks:
        nonempty = stack.mask.flatten(1).any(1)
        stack.slices /= torch.quantile(stack.slices[stack.mask], 0.99)
        slices.extend(stack[nonempty])
    return slices


def _setup(args: Namespace) -> None:
    level = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}[args.verbose]
    handlers = [logging.StreamHandler(sys.stderr)]
    if args.output_log:
        handlers.append(logging.FileHandler(args.output_log, mode="w"))
    logging.basicConfig(level=level, format="%(asctime)s %(levelname)s %(message)s", handlers=handlers, force=True)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    args.device = torch.device("cuda", args.device)
    torch.cuda.set_device(args.device)


def _load_model(args: Namespace):
    from .image_io import load_model

    inr, mask, stored = load_model(args.input_model, args.device)
    return inr, mask, merge_args(stored, args)


def _outputs(data: Dict, args: Namespace) -> None:
    """cli/io.py:33-50"""
    from .image_io import save_model, save_slices

    if getattr(args, "output_volume", None) and "output_volume" in data:
        if args.output_intensity_mean:
            data["output_volume"].rescale(args.output_intensity_mean)
        data["output_volume"].save(args.output_volume)
    if getattr(args, "output_model", None) and "output_model" in data:
        save_model(args.output_model, data["output_model"], data["mask"], args)
    for key in ("output_slices", "simulated_slices"):
        if getattr(args, key, None) and key in data:
            os.makedirs(getattr(args, key), exist_ok=True)
            save_slices(getattr(args, key), data[key])


def _load_stacks(args: Namespace) -> List:
    from .image_io import load_stack

    for name in ("stack_masks", "thicknesses"):
        if getattr(args, name, None) is not None and len(getattr(args, name)) != len(args.input_stacks):
            raise SystemExit(f"The numbers of {name.replace('_', ' ')} and input stacks are different!")
    stacks = []
    for i, f in enumerate(args.input_stacks):
        st = load_stack(f, args.stack_masks[i] if args.stack_masks is not None else None, device=args.device)
        if args.thicknesses is not None:
            st.thickness = args.thicknesses[i]
        stacks.append(st)
    return stacks


def register(args: Namespace, stacks: List) -> List:
    """Stacks -> motion-corrected slices (cli/commands.py:171-176): ``none`` keeps the nominal poses, ``stack``
    registers every stack to the first one; the SVoRT-based choices need the pretrained transformer (out of scope)."""
    if args.registration not in ("none", "stack"):
        raise NotImplementedError(f"--registration {args.registration}: the SVoRT transformer is out of scope of this "
                                  "build; register with the reference and pass the result through --input-slices, or use "
                                  "--registration stack / none")
    if args.registration == "stack":
        from .registration import register_stacks

        t1 = time.time()
        stacks = register_stacks(stacks)
        logging.info("Stack registration finished in %.1f s", time.time() - t1)
    return stacks_to_slices(stacks)


def register_cmd(args: Namespace) -> None:
    _outputs({"output_slices": register(args, _load_stacks(args))}, args)


def reconstruct(args: Namespace) -> None:
    from .image_io import load_slices, load_stack
    from .sample import sample_slices, sample_volume
    from .train import train

    if args.input_slices is None and args.input_stacks is None:
        raise SystemExit("No image data provided! Use --input-slices or --input-stacks to input data.")
    if args.input_slices is not None and (args.input_stacks or args.stack_masks or args.thicknesses):
        logging.warning("Since <input-slices> is provided, <input-stacks>, <stack_masks> and <thicknesses> would be ignored.")
        args.input_stacks = args.stack_masks = args.thicknesses = None
    for name in ("stack_masks", "thicknesses"):
        if getattr(args, name) is not None and len(getattr(args, name)) != len(args.input_stacks):
            raise SystemExit(f"The numbers of {name.replace('_', ' ')} and input stacks are different!")
    if args.output_volume is None and args.output_model is None:
        logging.warning("Both <output-volume> and <output-model> are not provided.")
    if not args.inference_batch_size:
        args.inference_batch_size = 8 * args.batch_size
    if not args.n_inference_samples:
        args.n_inference_samples = 2 * args.n_samples
    args.dtype = torch.float32 if args.single_precision else torch.float16
    if args.mlp_bf16 and not args.single_precision:
        raise SystemExit("--mlp-bf16 is a variant of the fp32 model: pass --single-precision too")
    if getattr(args, "mlp_fp16", False) and not args.single_precision:
        raise SystemExit("--mlp-fp16 is a variant of the fp32 model: pass --single-precision too")
    if getattr(args, "fp16_loss_scaling", False) and args.single_precision:
        raise SystemExit("--fp16-loss-scaling is the reference's DEFAULT numerics (fp16 operands + GradScaler): drop --single-precision")
    if not args.single_precision:
        logging.info("half-precision model structure (bias-free networks): %s matrix operands, fp32 accumulation",
                     "fp16 (with the reference's loss scaler: init 1, growth 2 / 2000 steps, backoff 0.5)" if getattr(args, "fp16_loss_scaling", False) else "bf16")
    t0 = time.time()
    if args.input_slices is not None:
        slices = load_slices(args.input_slices, args.device)
    else:
        slices = register(args, _load_stacks(args))
    logging.info("Data loading finished in %.1f s (%d slices)", time.time() - t0, len(slices))
    t0 = time.time()
    model, output_slices, mask = train(slices, args)
    logging.info("Reconstruction finished in %.1f s", time.time() - t0)
    t0 = time.time()
    data = {"mask": mask, "output_model": model, "output_slices": output_slices}
    if args.output_volume:
        data["output_volume"] = sample_volume(model, mask, args)
    if args.simulated_slices:
        data["simulated_slices"] = sample_slices(model, output_slices, mask, args)
    _outputs(data, args)
    logging.info("Results saving finished in %.1f s", time.time() - t0)


def sample_volume_cmd(args: Namespace) -> None:
    from .sample import sample_volume

    model, mask, args = _load_model(args)
    if not getattr(args, "inference_batch_size", None):
        args.inference_batch_size = 8 * args.batch_size
    if not getattr(args, "n_inference_samples", None):
        args.n_inference_samples = 2 * args.n_samples
    _outputs({"output_volume": sample_volume(model, mask, args)}, args)


def sample_slices_cmd(args: Namespace) -> None:
    from .image_io import load_slices
    from .sample import sample_slices

    model, mask, args = _load_model(args)
    slices = load_slices(args.input_slices, args.device)
    _outputs({"simulated_slices": sample_slices(model, slices, mask, args)}, args)


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    _setup(args)
    t0 = time.time()
    {"reconstruct": reconstruct, "register": register_cmd, "sample-volume": sample_volume_cmd,
     "sample-slices": sample_slices_cmd}[args.command](args)
    logging.info("Command 'nesvor %s' finished, overall time: %.1f s", args.command, time.time() - t0)


if __name__ == "__main__":
    main()
