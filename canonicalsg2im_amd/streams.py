"""Side streams for the small, latency-bound pieces of the training step (round 5).

The scene-graph encoder (~150 small dependent launches) and the object-crop discriminator's passes share nothing with the
generator's / image discriminator's convolutions that run next to them (the generator consumes the ground-truth boxes:
sg2im/meta_models.py:47 of the reference), so they are issued on a stream of their own and joined where their results are
first needed.  Autograd replays a backward node on the stream its forward ran on and orders the streams itself, so the
backward of such a piece overlaps too.  `CSG_EAGER_OVERLAP=0`: everything on the current stream."""
import contextlib
import os

import torch

ENABLED = os.environ.get("CSG_EAGER_OVERLAP", "1") != "0"
_STREAMS = {}


def usable(*tensors):
    """Overlap only for HIP tensors outside a graph capture (a capture has its own stream discipline: graphs.py)."""
    return (ENABLED and all(t is None or t.is_cuda for t in tensors) and any(t is not None for t in tensors)
            and not torch.cuda.is_current_stream_capturing())


@contextlib.contextmanager
def beside(tag, device):
    """Run the body on side stream `tag` of `device`, ordered behind everything the current stream holds so far."""
    key = (tag, torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
    side = _STREAMS.get(key)
    if side is None:
        side = _STREAMS[key] = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        yield side


def join(side, *tensors):
    """The current stream waits for `side`; `tensors` (produced there, consumed here) are recorded for the caching
    allocator, which would otherwise hand their blocks back to the side stream while this stream still reads them."""
    main = torch.cuda.current_stream(side.device)
    main.wait_stream(side)
    for t in tensors:
        if torch.is_tensor(t):
            t.record_stream(main)
        elif isinstance(t, dict):
            for v in t.values():
                if torch.is_tensor(v):
                    v.record_stream(main)
        elif isinstance(t, (list, tuple)):
            join_records(main, t)


def join_records(main, items):
    for v in items:
        if torch.is_tensor(v):
            v.record_stream(main)
        elif isinstance(v, (list, tuple)):
            join_records(main, v)
