"""Batch contract + step helpers (mirror of asr_deepspeech/functional.py)."""
import math

import torch
import torch.distributed as dist


def to_np(x):
    return x.cpu().numpy()


def _collate_fn(batch):
    """functional.py:9-32 — sort by frame count (descending), zero-pad to (B,1,F,Tmax), emit
    percentages (float32 of python-float T_b/Tmax), flat int32 targets in sorted order, target sizes."""
    batch = sorted(batch, key=lambda sample: sample[0].size(1), reverse=True)
    longest = batch[0][0]
    freq_size, max_len = longest.size(0), longest.size(1)
    n = len(batch)
    inputs = torch.zeros(n, 1, freq_size, max_len)
    input_percentages = torch.zeros(n, dtype=torch.float32)
    target_sizes = torch.zeros(n, dtype=torch.int32)
    targets = []
    for i, (spect, target) in enumerate(batch):
        t = spect.size(1)
        inputs[i, 0, :, :t] = spect
        input_percentages[i] = t / float(max_len)
        target_sizes[i] = len(target)
        targets.extend(target)
    return inputs, torch.tensor(targets, dtype=torch.int32), input_percentages, target_sizes


def reduce_tensor(tensor, world_size, reduce_op_max=False):
    """functional.py:35-42 (dead code in the reference; live here for metric averaging over RCCL)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.MAX if reduce_op_max else dist.ReduceOp.SUM)
    if not reduce_op_max:
        rt /= world_size
    return rt


def check_loss(loss, loss_value):
    """functional.py:45-61 — a loss is invalid if +-inf, NaN or negative; returns (valid, message)."""
    if loss_value == float("inf") or loss_value == float("-inf"):
        return False, "WARNING: received an inf loss"
    if math.isnan(loss_value) or (torch.is_tensor(loss) and bool(torch.isnan(loss).sum() > 0)):
        return False, "WARNING: received a nan loss, setting loss value to 0"
    if loss_value < 0:
        return False, "WARNING: received a negative loss"
    return True, ""
