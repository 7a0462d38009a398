"""Small host helpers with the reference's names (reference utils.py:20-24, 187-192)."""
import torch


def get_mask_from_lengths(lengths):
    """True where position < length (device-agnostic; the reference hard-codes torch.cuda.LongTensor)."""
    max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device)
    return ids < lengths.unsqueeze(1)


def to_gpu(x):
    x = x.contiguous()
    if torch.cuda.is_available():
        x = x.cuda(non_blocking=True)
    return x
