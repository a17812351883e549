"""Deterministic model weights shared by oracle/gen_golden.py (reference side) and the GPU tests (this package's side)."""
def seeded_parameters_(model, std: float = 0.05, strip: str = ""):
    """Deterministic weights WITHOUT a stored state dict: every parameter is drawn from its own CPU generator seeded with the crc32
    of its (stripped) name -- the same call on the reference's model (oracle/gen_golden.py) and on this package's model gives the
    same tensors, so a BASELINE-sized layer can be pinned by a fixture that holds only inputs, ranges and outputs."""
    import zlib
    import torch
    with torch.no_grad():
        for name, p in model.named_parameters():
            key = name[len(strip):] if strip and name.startswith(strip) else name
            g = torch.Generator(device="cpu").manual_seed(zlib.crc32(key.encode()))
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * std)
            else:
                p.copy_(1.0 + 0.3 * (torch.rand(p.shape, generator=g) - 0.5))
    return model
