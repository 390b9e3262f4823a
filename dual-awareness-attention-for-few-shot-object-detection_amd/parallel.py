"""Multi-GPU sharding of the episode batch: one process per GPU, every (query, k-shot
support) episode is independent in the forward path (all BN frozen, dana.py:379-385), so ranks take
disjoint slices of the five input tensors and there is NO data-path collective; the only exchange
step is the timing/metric reduction of bench.py and, for training, the gradient all-reduce (the
reference's nn.DataParallel reduce-add at train.py:104-105,138-139 becomes an RCCL allReduce / world).
Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo"."""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous, balanced [begin, end) of `total` episodes for `rank` (the first total % world ranks
    get one extra) -- the same split nn.DataParallel's scatter produces on dim 0."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_episode(inputs, rank, world):
    """inputs = (im_data, im_info, gt_boxes, num_boxes, support_ims): slice dim 0 of each for `rank`."""
    b0, b1 = shard_bounds(inputs[0].size(0), rank, world)
    return tuple(t[b0:b1] for t in inputs)


def allreduce_mean_(tensors, world=None, bucket_bytes=25 << 20):
    """Bucketed in-place mean all-reduce (gradients of the 70 trainable tensors, 148.5 MB fp32): tensors
    are packed into ~25 MB flat buckets so each collective is large enough for the per-link xGMI ring."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tensors
    world = world or dist.get_world_size()
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        bucket.clear()

    for t in tensors:
        bucket.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
            size = 0
    flush()
    return tensors


def max_over_ranks(value, device):
    """bench.py contract: the step time is the MAX over ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
