"""Frame sharding of a scan stream over the GPUs of one node, and the fixed-stride keypoint
records that are gathered with one RCCL collective per batch.

Scans are independent (the reference keeps no state across scans except roll/pitch, which are
per-scan inputs: ref node.h:116), so rank r simply owns a contiguous block of the stream and
runs the unchanged single-GPU pipeline on it; the only exchange is the keypoint table.
What crosses GPUs (since 0.7) is ONE compact keypoint block per rank and batch — fx_pack_keypoint_block:
row 0 {scans, keypoints stored, OR of the flags, max_total} (u32), kp_offset[max_scans + 1], flags[max_scans],
then max_total (x, y, z, elevation) rows packed in scan order (block_rows / pack_block / unpack_block below) —
a quarter of the bytes of the fixed-stride records it replaces: a VLP-16 scan has 54 keypoints where a record
reserves the context's capacity (256).  The fixed-stride record of one scan = (1 + rec_kp) float4: header
{n_kp, flags, 0, 0} as uint32, then rec_kp (x, y, z, elevation) entries, zero padded — fx_pack_keypoint_records —
stays in the C-ABI (a consumer that wants random access by scan without the offsets).
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


def block_size(total, world):
    """Records every rank contributes to the all-gather of a stream of `total` scans: the largest block of the plan (a
    collective takes equal contributions; the ranks whose block is one scan shorter pad theirs with an empty record)."""
    return max(shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)) if world else 0


def pad_block(rec, total, world):
    """A rank's [b, 1 + rec_kp, 4] records padded with empty records (n_kp = 0) to block_size(total, world)."""
    bs = block_size(total, world)
    if rec.shape[0] == bs:
        return rec
    out = np.zeros((bs,) + rec.shape[1:], rec.dtype)
    out[:rec.shape[0]] = rec
    return out


def stream_order(table, total, world):
    """The gathered table ([world * block_size, 1 + rec_kp, 4], rank blocks one after the other, padded) -> the `total`
    records of the stream in stream order (the padding dropped)."""
    bs = block_size(total, world)
    rows = [r * bs + i for r in range(world) for i in range(shard_range(total, world, r)[1] - shard_range(total, world, r)[0])]
    return table[rows]


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


FX_FLAG_KP_OVERFLOW = 0x4


def block_rows(max_scans, max_total):
    """float4 rows of a compact keypoint block (fx_keypoint_block_bytes / 16)."""
    return 1 + (max_scans + 1 + 3) // 4 + (max_scans + 3) // 4 + max_total


def block_keypoints_per_scan(max_scans, per_scan=64):
    """The keypoint capacity bench.py and fx::MultiGpu give a block: `per_scan` keypoints a scan of the batch on average
    (VLP-16 scenes have 54; what does not fit is flagged, never silent)."""
    return max_scans * per_scan


def pack_block(keypoints_per_scan, flags_per_scan, max_scans, max_total):
    """Host-side statement of fx_pack_keypoint_block (used by the CPU tests): [block_rows, 4] float32."""
    nb = min(len(keypoints_per_scan), max_scans)
    blk = np.zeros((block_rows(max_scans, max_total), 4), np.float32)
    u = blk.view(np.uint32).reshape(-1)
    n_off = 4 * ((max_scans + 1 + 3) // 4)
    off = np.zeros(n_off, np.int64)
    run = 0
    true_off = [0]
    for b in range(nb):
        run += len(keypoints_per_scan[b])
        true_off.append(run)
    for i in range(n_off):
        off[i] = min(true_off[min(i, nb)], max_total)
    u[4:4 + n_off] = off
    f0 = 4 + n_off
    flags_or = FX_FLAG_KP_OVERFLOW if len(keypoints_per_scan) > max_scans else 0
    for b in range(nb):
        kept = min(true_off[b + 1], max_total) - min(true_off[b], max_total)
        f = int(flags_per_scan[b]) | (FX_FLAG_KP_OVERFLOW if kept < len(keypoints_per_scan[b]) else 0)
        u[f0 + b] = f
        flags_or |= f
    k0 = 1 + n_off // 4 + (max_scans + 3) // 4
    for b in range(nb):
        n = int(off[b + 1] - off[b])
        if n:
            blk[k0 + off[b]:k0 + off[b] + n] = np.asarray(keypoints_per_scan[b], np.float32)[:n]
    u[0:4] = (nb, min(true_off[nb], max_total), flags_or, max_total)
    return blk


def unpack_block(blk, max_scans):
    """(header dict, [(n_kp, flags, keypoints[n_kp, 4])] for the block's scans) from one compact block ([rows, 4] float32)."""
    blk = np.ascontiguousarray(blk, dtype=np.float32).reshape(-1, 4)
    u = blk.view(np.uint32).reshape(-1)
    nb, total, flags_or, max_total = (int(v) for v in u[0:4])
    n_off = 4 * ((max_scans + 1 + 3) // 4)
    off = u[4:4 + n_off].astype(np.int64)
    f0 = 4 + n_off
    k0 = 1 + n_off // 4 + (max_scans + 3) // 4
    assert blk.shape[0] == block_rows(max_scans, max_total), (blk.shape, max_scans, max_total)
    out = []
    for b in range(nb):
        n = int(off[b + 1] - off[b])
        out.append((n, int(u[f0 + b]), blk[k0 + off[b]:k0 + off[b] + n].copy()))
    return dict(scans=nb, keypoints=total, flags_or=flags_or, max_total=max_total), out


def stream_order_blocks(table, total, world, max_scans):
    """The gathered blocks ([world * block_rows, 4]: rank blocks one after the other) -> the `total` scans of the stream in
    stream order as (n_kp, flags, keypoints) — rank r's block holds the scans of shard_range(total, world, r)."""
    table = np.ascontiguousarray(table, dtype=np.float32).reshape(world, -1, 4)
    out = []
    for r in range(world):
        lo, hi = shard_range(total, world, r)
        hdr, scans = unpack_block(table[r], max_scans)
        assert hdr["scans"] == hi - lo, (r, hdr, lo, hi)
        out.extend(scans)
    return out


def gather_records_to_root(rec_tensor, world, root=0, out=None):
    """The gather proper (north_star: "an RCCL gather of keypoints"): only `root` receives the table — a publisher needs one
    copy, and the other ranks' HBM and xGMI ingress stay out of it.  torch.distributed form (gloo test; bench.py's RCCL form is
    RcclGather.gather).  Returns the [world * n] table on root, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return rec_tensor
    rank = dist.get_rank()
    flat = rec_tensor.contiguous().view(-1)
    parts = None
    if rank == root:
        if out is None:
            out = torch.empty((world * flat.numel(),), dtype=flat.dtype, device=flat.device)
        parts = list(out.view(world, -1).unbind(0))
    dist.gather(flat, gather_list=parts, dst=root)
    return out.view((world * rec_tensor.shape[0],) + tuple(rec_tensor.shape[1:])) if rank == root else None


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


class RcclGather:
    """The path's one collective, issued straight on the stream the batch runs on: ncclAllGather through RCCL's C API
    (the librccl.so torch ships, so the process keeps ONE RCCL).  torch.distributed brings the ranks up and carries the
    unique ids; the collective itself is then an ordinary kernel of the context's own stream — no side stream, no
    cross-stream events (through torch.distributed's stream juggling the same gather cost 19 % of the throughput with
    three batches in flight, this way 2 %).  bench.py uses ONE communicator for all contexts in flight and issues the
    gathers in step order on every rank — RCCL's ordering contract; several communicators per device (n_comms > 1) are
    only safe when their kernels can always run side by side.  csrc/fx_multi.hpp does the same from C++."""

    @staticmethod
    def _load():
        import ctypes as C
        import os
        import torch
        lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclGather", "ncclCommDestroy", "ncclGetErrorString"):
            getattr(lib, sym)
        return lib

    @staticmethod
    def available():
        """True when RCCL's C API can be loaded in this process.  Every rank checks this (and the ranks agree on the
        answer through a collective) BEFORE any communicator is created: a rank that failed later would leave the
        others waiting inside ncclCommInitRank."""
        try:
            RcclGather._load()
            return True
        except (OSError, AttributeError):
            return False

    def __init__(self, world, rank, device, n_comms=1):
        import ctypes as C
        import torch
        import torch.distributed as dist
        self.C, self.world, self.rank = C, world, rank
        self.lib = self._load()
        self.lib.ncclGetErrorString.restype = C.c_char_p

        class _Uid(C.Structure):  # ncclUniqueId is passed by value
            _fields_ = [("internal", C.c_byte * 128)]
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
        self.lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.ncclGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
        torch.cuda.set_device(device)
        self.comms = []
        for _ in range(n_comms):
            u = _Uid()
            if rank == 0:
                self._check(self.lib.ncclGetUniqueId(C.byref(u)))
            t = torch.frombuffer(bytearray(bytes(u)), dtype=torch.uint8).clone()
            if dist.is_available() and dist.is_initialized() and world > 1:
                t = t.to(device)
                dist.broadcast(t, 0)
                t = t.cpu()
            C.memmove(C.byref(u), t.numpy().tobytes(), 128)
            comm = C.c_void_p()
            self._check(self.lib.ncclCommInitRank(C.byref(comm), world, u, rank))
            self.comms.append(comm)

    def _check(self, r):
        if r != 0:
            raise RuntimeError("RCCL: " + self.lib.ncclGetErrorString(r).decode())

    def all_gather(self, rec_tensor, out_tensor, stream_ptr, comm=0):
        """Every rank's record block into out_tensor ([world * B, 1 + rec_kp, 4]), in rank (= stream) order, enqueued
        on the HIP stream `stream_ptr` (the context's): ordered behind fx_pack_keypoint_records like any kernel."""
        assert out_tensor.numel() == self.world * rec_tensor.numel()
        self._check(self.lib.ncclAllGather(rec_tensor.data_ptr(), out_tensor.data_ptr(), rec_tensor.numel(), 7,  # ncclFloat32
                                           self.comms[comm], self.C.c_void_p(stream_ptr)))

    def gather(self, rec_tensor, out_tensor, stream_ptr, root=0, comm=0):
        """The same to `root` only (ncclGather): out_tensor is written on root and may be None elsewhere."""
        if self.rank == root:
            assert out_tensor is not None and out_tensor.numel() == self.world * rec_tensor.numel()
        self._check(self.lib.ncclGather(rec_tensor.data_ptr(), out_tensor.data_ptr() if out_tensor is not None else None, rec_tensor.numel(), 7,
                                        root, self.comms[comm], self.C.c_void_p(stream_ptr)))

    def close(self):
        for c in self.comms:
            self.lib.ncclCommDestroy(c)
        self.comms = []
