"""The only collective of the path: summing the per-rank float framebuffers onto rank 0.

Ranks own disjoint image tiles (pearray_amd.tiling), so no data-path exchange happens while rendering; the
reference's equivalent is the offline `pr_imagemerge.py` sum of per-tile EXRs (tools/pr_imagemerge.py) and the
mutex-protected FrameOutputDevice::mergeLocal inside one process.  On MI355X nodes the reduce runs over
RCCL/xGMI (torch.distributed backend "nccl"); CPU tests use gloo.
"""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def reduce_framebuffer(xyz, samples, dst=0):
    """In-place sum onto rank `dst`: xyz float32 [H,W,3], samples int32 [H,W]."""
    if world()[1] == 1:
        return
    if dist.get_backend() == "gloo" and xyz.is_cuda:
        # rehearsal on a one-GPU box (bench.py --rehearse-on-one-gpu): stage through host memory
        hx, hs = xyz.cpu(), samples.cpu()
        dist.reduce(hx, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(hs, dst=dst, op=dist.ReduceOp.SUM)
        xyz.copy_(hx)
        samples.copy_(hs)
        return
    dist.reduce(xyz, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(samples, dst=dst, op=dist.ReduceOp.SUM)


def _scalar(value, op, device=None):
    if world()[1] == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=op)
    return float(t.item())


def sum_scalar(value, device=None):
    return _scalar(value, dist.ReduceOp.SUM, device)


def max_scalar(value, device=None):
    return _scalar(value, dist.ReduceOp.MAX, device)


def gather_scalars(value, device=None):
    """One float per rank, in rank order, on every rank (diagnostics: which rank was slowest)."""
    n = world()[1]
    if n == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    out = [torch.zeros_like(t) for _ in range(n)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
