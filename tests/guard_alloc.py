"""Guard bands around every device buffer the host code hands to a kernel as an OUTPUT (test infrastructure).

``with GuardedAllocations() as g:`` replaces torch.empty / zeros / full / empty_like / zeros_like for CUDA tensors by views
into larger byte buffers whose margins hold a sentinel pattern; ``g.check()`` names every allocation whose margins were
written.  A kernel that stores a few elements past its output is harmless while whatever lies behind it happens to be
dead memory of the same stream -- and a silent, timing-dependent corruption as soon as a second stream (a sweep preparing
the next target, Docker.prepare) has live data there."""
import traceback

import torch

PAD = 1 << 20                # bytes on either side
SENTINEL = 0xA5


class GuardedAllocations(object):
    NAMES = ("empty", "zeros", "full", "empty_like", "zeros_like")

    def __init__(self, min_bytes=1024, empty_byte=SENTINEL):
        """empty_byte: what an ``empty`` buffer holds before its first kernel (0xFF: float NaNs -- a result that depends
        on memory nobody wrote shows up as a changed or NaN result between two values of this)."""
        self.items, self.min_bytes, self.empty_byte = [], min_bytes, empty_byte
        self.orig = {n: getattr(torch, n) for n in self.NAMES}

    # ---- the replacements ------------------------------------------------------------------------------------------
    def _guarded(self, shape, dtype, device, fill):
        dtype = dtype or torch.get_default_dtype()
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        body = (nbytes + 255) // 256 * 256
        raw = self.orig["full"]((PAD + body + PAD,), SENTINEL, dtype=torch.uint8, device=device)
        if fill is None and self.empty_byte != SENTINEL:
            raw[PAD:PAD + nbytes] = self.empty_byte
        view = raw[PAD:PAD + nbytes].view(dtype).view(tuple(int(s) for s in shape))
        if fill is not None:
            view.fill_(fill)
        where = [f for f in traceback.extract_stack(limit=8) if "guard_alloc" not in f.filename][-1]
        self.items.append((raw, nbytes, "%s:%d %s %s" % (where.filename.split("/")[-1], where.lineno, tuple(shape), dtype)))
        return view

    def _wants(self, device, shape, dtype):
        if device is None or torch.device(device).type != "cuda":
            return False
        n = 1
        for s in shape:
            n *= int(s)
        return n * 4 >= self.min_bytes

    @staticmethod
    def _shape(args):
        if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
            return tuple(args[0])
        return tuple(args)

    def __enter__(self):
        g = self

        def empty(*args, dtype=None, device=None, **kw):
            shape = g._shape(args)
            return g._guarded(shape, dtype, device, None) if g._wants(device, shape, dtype) and not kw else g.orig["empty"](*args, dtype=dtype, device=device, **kw)

        def zeros(*args, dtype=None, device=None, **kw):
            shape = g._shape(args)
            return g._guarded(shape, dtype, device, 0) if g._wants(device, shape, dtype) and not kw else g.orig["zeros"](*args, dtype=dtype, device=device, **kw)

        def full(size, fill_value, dtype=None, device=None, **kw):
            shape = tuple(size)
            if g._wants(device, shape, dtype) and not kw and dtype is not None:
                return g._guarded(shape, dtype, device, fill_value)
            return g.orig["full"](size, fill_value, dtype=dtype, device=device, **kw)

        def empty_like(t, **kw):
            return g._guarded(tuple(t.shape), t.dtype, t.device, None) if (t.is_cuda and not kw and t.numel() * 4 >= g.min_bytes) else g.orig["empty_like"](t, **kw)

        def zeros_like(t, **kw):
            return g._guarded(tuple(t.shape), t.dtype, t.device, 0) if (t.is_cuda and not kw and t.numel() * 4 >= g.min_bytes) else g.orig["zeros_like"](t, **kw)

        for n, f in (("empty", empty), ("zeros", zeros), ("full", full), ("empty_like", empty_like), ("zeros_like", zeros_like)):
            setattr(torch, n, f)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(torch, n, f)
        return False

    # ---- the verdict -----------------------------------------------------------------------------------------------
    def check(self, release=True):
        """-> list of 'where: N bytes written below / above' for every violated allocation."""
        torch.cuda.synchronize()
        bad = []
        for raw, nbytes, where in self.items:
            body = (nbytes + 255) // 256 * 256
            lo = int((raw[:PAD] != SENTINEL).sum())
            hi = int((raw[PAD + body:] != SENTINEL).sum())
            # (the round-up slack [nbytes, body) belongs to nobody either)
            slack = int((raw[PAD + nbytes:PAD + body] != SENTINEL).sum()) if body > nbytes else 0
            if lo or hi or slack:
                first_hi = int((raw[PAD + nbytes:] != SENTINEL).nonzero()[0]) if (hi or slack) else -1
                bad.append("%s: %d bytes written below, %d above (first at +%d past the end)" % (where, lo, hi + slack, first_hi))
        if release:
            self.items = []
        return bad
