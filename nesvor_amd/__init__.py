from . import ops  # noqa: F401  (registers the torch.ops.nesvor dispatcher ops)
