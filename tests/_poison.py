"""Debugging aid for the -m gpu tests (not a test): W3D_TEST_FILL=zeros|ff|nan|small|rand|unit makes every torch.empty /
empty_like / new_empty on the GPU come back FILLED with that pattern instead of whatever the caching allocator hands out, so
that a kernel reading memory it (or its predecessor) never wrote shows up as a reproducible failure instead of one that
depends on what the previous owner of the block left behind:
  zeros  all bytes 0 (what a fresh box's memory mostly holds)      ff    all bytes 0xFF
  nan    0x7FC00000 words                                          small random int32 in [0, 64)
  rand   random int32 bit patterns                                 unit  random floats in [0, 1)
The product never imports this; tests/conftest.py and the two-rank worker install it when the variable is set."""
import os


def install(mode):
    import torch
    if getattr(torch, "_w3d_poisoned", None):
        return
    torch._w3d_poisoned = mode
    orig_empty, orig_empty_like = torch.empty, torch.empty_like
    orig_new_empty = torch.Tensor.new_empty

    def fill(t):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.numel() and t.is_contiguous()):
            return t
        nbytes = t.numel() * t.element_size()
        with torch.no_grad():
            raw = t.view(-1).view(torch.uint8)
            if mode == "zeros":
                raw.zero_()
            elif mode == "ff":
                raw.fill_(255)
            else:
                n4 = nbytes // 4
                if n4:
                    w = raw[:4 * n4].view(torch.int32)
                    if mode == "nan":
                        w.fill_(0x7FC00000)
                    elif mode == "small":
                        w.random_(0, 64)
                    elif mode == "rand":
                        w.random_(-2 ** 31, 2 ** 31 - 1)
                    elif mode == "unit":
                        raw[:4 * n4].view(torch.float32).uniform_(0.0, 1.0)
                    else:
                        raise ValueError(f"W3D_TEST_FILL={mode}")
                if nbytes % 4:
                    raw[4 * n4:].fill_(0 if mode in ("small", "unit") else 255)
        return t

    def empty(*a, **k):
        return fill(orig_empty(*a, **k))

    def empty_like(*a, **k):
        return fill(orig_empty_like(*a, **k))

    def new_empty(self, *a, **k):
        return fill(orig_new_empty(self, *a, **k))
    torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty


def install_from_env():
    mode = os.environ.get("W3D_TEST_FILL", "")
    if mode:
        install(mode)
