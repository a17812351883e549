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


_PI_SEED = 4242


def bigram_permutation(vocab: int):
    """The fixed vocabulary permutation pi of the contractive fixture model: the token after t is pi(t)."""
    import torch
    return torch.randperm(vocab, generator=torch.Generator(device="cpu").manual_seed(_PI_SEED))


def bigram_sequence(vocab: int, length: int, generator, follow: float = 0.8):
    """[1, length] token ids: the next token is pi(previous) with probability `follow`, else uniform -- text the contractive model
    predicts the way a small LM predicts real text (perplexity ~13), every logit counting."""
    import torch
    pi = bigram_permutation(vocab)
    jump = torch.rand(length, generator=generator) >= follow
    rand = torch.randint(0, vocab, (length,), generator=generator)
    ids = torch.empty(length, dtype=torch.long)
    ids[0] = rand[0]
    for t in range(1, length):
        ids[t] = rand[t] if jump[t] else pi[ids[t - 1]]
    return ids.view(1, -1)


def seeded_contractive_parameters_(model, strip: str = "", beta: float = 12.0, branch_o: float = 1.25, branch_w2: float = 0.9):
    """Deterministic weights of a CONTRACTIVE decoder (oracle/gen_golden.py: gen_full_depth_stable_case; the GPU tests rebuild them):
    what a trained checkpoint has and seeded_parameters_' random model lacks.  Unit-variance embeddings own the residual stream;
    o_proj / w2 are scaled down so a branch adds ~0.25 of the stream's RMS (a perturbation is carried along, not amplified); the
    other projections decay row-wise (singular values fall off instead of sitting on a Marchenko-Pastur bulk); the unembedding row of
    pi(t) is beta / hidden x the embedding of t -- a peaked next-token distribution.  Drawn per parameter from a generator seeded with
    the crc32 of its (stripped) name, like seeded_parameters_."""
    import zlib
    import torch
    with torch.no_grad():
        named = dict(model.named_parameters())
        embed = None
        for name, p in named.items():
            key = name[len(strip):] if strip and name.startswith(strip) else name
            g = torch.Generator(device="cpu").manual_seed(zlib.crc32(("contractive/" + key).encode()))
            if key.endswith("lm_head.weight"):
                continue
            if p.dim() < 2:
                p.copy_(1.0 + 0.3 * (torch.rand(p.shape, generator=g) - 0.5))
            elif "embed_tokens" in key:
                p.copy_(torch.randn(p.shape, generator=g))
                embed = p
            else:
                w = torch.randn(p.shape, generator=g) * 0.02
                if "o_proj" in key:
                    w *= branch_o
                elif key.endswith("w2.weight"):
                    w *= branch_w2
                else:
                    rows = p.shape[0]
                    w *= torch.rsqrt(1.0 + 3.0 * torch.arange(rows, dtype=torch.float32) / rows).view(-1, 1)
                p.copy_(w)
        head = next(p for n, p in named.items() if n.endswith("lm_head.weight"))
        vocab, hidden = head.shape
        inv = torch.empty(vocab, dtype=torch.long)
        inv[bigram_permutation(vocab)] = torch.arange(vocab)
        head.copy_(embed.detach().float()[inv] * (beta / hidden))
    return model
