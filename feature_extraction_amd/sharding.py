"""Frame sharding of a scan stream over the GPUs of one node, and the fixed-stride keypoint
records that are gathered with one RCCL collective per batch.

Scans are independent (the reference keeps no state across scans except roll/pitch, which are
per-scan inputs: ref node.h:116), so rank r simply owns a contiguous block of the stream and
runs the unchanged single-GPU pipeline on it; the only exchange is the keypoint table.
Record of one scan = (1 + rec_kp) float4: header {n_kp, flags, 0, 0} as uint32, then rec_kp
(x, y, z, elevation) entries, zero padded — what fx_pack_keypoint_records writes on the device.
"""
import numpy as np

REC_KP = 127  # 1 header + 127 keypoints = 2 KiB per scan


def shard_range(total, world, rank):
    """Contiguous block [start, end) of a stream of `total` scans owned by `rank` of `world`."""
    return total * rank // world, total * (rank + 1) // world


def owner_of(scan, total, world):
    """Rank that owns stream position `scan` under shard_range."""
    r = (scan * world) // total if total else 0
    while r + 1 < world and scan >= shard_range(total, world, r)[1]:
        r += 1
    while r > 0 and scan < shard_range(total, world, r)[0]:
        r -= 1
    return r


def pack_records(keypoints_per_scan, flags_per_scan, rec_kp=REC_KP):
    """Host-side statement of fx_pack_keypoint_records (used by the CPU tests)."""
    B = len(keypoints_per_scan)
    rec = np.zeros((B, 1 + rec_kp, 4), np.float32)
    hdr = rec.view(np.uint32)
    for b, kp in enumerate(keypoints_per_scan):
        k = min(len(kp), rec_kp)
        hdr[b, 0, 0] = k
        hdr[b, 0, 1] = int(flags_per_scan[b]) | (0x4 if len(kp) > rec_kp else 0)
        rec[b, 1:1 + k] = kp[:k]
    return rec


def unpack_records(rec):
    """[(n_kp, flags, keypoints[n_kp, 4])] from a gathered record table."""
    rec = np.ascontiguousarray(rec, dtype=np.float32)
    hdr = rec.view(np.uint32)
    out = []
    for b in range(rec.shape[0]):
        k = int(hdr[b, 0, 0])
        out.append((k, int(hdr[b, 0, 1]), rec[b, 1:1 + k].copy()))
    return out


def all_gather_records(rec_tensor, world, out=None, async_op=False):
    """One collective per batch: every rank's record block, in rank (= stream) order.
    out: preallocated [world * B, 1 + rec_kp, 4] table (steady state: no allocation); async_op: return
    (table, work) with the collective still in flight on its own stream (bench.py overlaps it with the
    kernels of the batches behind it).  This is the function both bench.py (RCCL) and the gloo test run."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return (rec_tensor, None) if async_op else rec_tensor
    if out is None:
        out = torch.empty((world * rec_tensor.shape[0],) + tuple(rec_tensor.shape[1:]), dtype=rec_tensor.dtype,
                          device=rec_tensor.device)
    work = dist.all_gather_into_tensor(out.view(-1), rec_tensor.contiguous().view(-1), async_op=async_op)
    return (out, work) if async_op else out
