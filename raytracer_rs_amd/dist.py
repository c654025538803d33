"""One frame over several GPUs of a node: row-band sharding and the gather of finished bands to rank 0.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" for rehearsal: bands then travel
through pinned host memory, gloo being a CPU backend).  The scene is
small and read-only, so every rank holds a copy; pixels are independent, so the only exchange is the gather of the
rendered rows.  The frame is cut into bands of `band_rows` rows and band b belongs to rank b % world_size: band cost
varies by an order of magnitude between sky and glass, and a fine round-robin interleave evens that out without any
run-time coordination between the processes.

The gather is `torch.distributed.gather`, which on the NCCL/RCCL backend is a group of point-to-point send/recv:
every peer's bands travel over that peer's own xGMI link into rank 0.  It runs on its own stream and overlaps the
rendering of the next frame (two send buffers per rank).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

DEFAULT_BAND_ROWS = 16


def n_bands(height: int, band_rows: int = DEFAULT_BAND_ROWS) -> int:
    return (height + band_rows - 1) // band_rows


def bands_of(rank: int, world_size: int, height: int, band_rows: int = DEFAULT_BAND_ROWS) -> List[int]:
    """Bands (indices) rank `rank` renders."""
    return list(range(rank, n_bands(height, band_rows), world_size))


def bands_per_rank(world_size: int, height: int, band_rows: int = DEFAULT_BAND_ROWS) -> int:
    """Band slots in every rank's send buffer (the gather needs equal sizes; ranks with one band fewer pad)."""
    return (n_bands(height, band_rows) + world_size - 1) // world_size


def band_rows_range(band: int, height: int, band_rows: int = DEFAULT_BAND_ROWS) -> Tuple[int, int]:
    return band * band_rows, min(height, (band + 1) * band_rows)


class BandGather:
    """Buffers and the collective for assembling a frame on rank 0.

    Every rank owns `depth` send buffers ([bands_per_rank, band_rows*width*channels] each; `buffer(k)` is the one
    frame k goes through, `mine` is buffer 0); rank 0 also owns `frame` ([n_bands, band_rows*width*channels], i.e.
    the frame padded to whole bands; `frame_rows()` views the real rows).

    On a GPU the gather of frame k runs on a stream of its own, so it overlaps the rendering of frame k+1 into the
    other send buffer: `begin_render(k)` makes the rendering stream wait until buffer(k) is free again,
    `gather(k)` hands the buffer over once everything queued on the rendering stream so far has finished, and
    `finish()` joins the two streams.  On the CPU (gloo rehearsal) the same calls run one after the other.
    """

    def __init__(self, width: int, height: int, rank: int, world_size: int, device: torch.device,
                 band_rows: int = DEFAULT_BAND_ROWS, channels: int = 3, dtype: torch.dtype = torch.float64,
                 depth: int = 2, collective_even_alone: bool = False):
        self.width, self.height, self.rank, self.world_size = width, height, rank, world_size
        self.band_rows, self.channels = band_rows, channels
        self.band_vals = band_rows * width * channels
        self.n_bands = n_bands(height, band_rows)
        self.slots = bands_per_rank(world_size, height, band_rows)
        # collective_even_alone: a process group of ONE rank still goes through the send buffers, the side stream and
        # the gather (the multi-rank path exercised end to end on a box with a single GPU)
        self.collective = world_size > 1 or collective_even_alone
        self.depth = max(1, depth) if self.collective else 1
        self.send = [torch.zeros(self.slots, self.band_vals, dtype=dtype, device=device) for _ in range(self.depth)]
        self.mine = self.send[0]
        self.frame: Optional[torch.Tensor] = None
        self._recv: Optional[List[torch.Tensor]] = None
        if rank == 0:
            # padded to slots * world_size bands, so that "band b belongs to rank b % world_size" is a plain reshape:
            # frame_padded[slot, rank] is band slot * world_size + rank
            self._frame_padded = torch.zeros(self.slots * world_size, self.band_vals, dtype=dtype, device=device)
            self.frame = self._frame_padded[: self.n_bands]
            if self.collective:
                # one tensor [world_size, slots, band_vals]; the gather receives into its slices, and one strided
                # copy puts every rank's bands in place (one kernel per frame, not one per rank)
                self._recv_all = torch.empty(world_size, self.slots, self.band_vals, dtype=dtype, device=device)
                self._recv = [self._recv_all[r] for r in range(world_size)]
        self._host_send: Optional[torch.Tensor] = None  # pinned staging for a gloo gather of device bands
        self._host_recv: Optional[List[torch.Tensor]] = None
        self._host_recv_all: Optional[torch.Tensor] = None
        self._comm = None
        if device.type == "cuda" and self.collective:
            self._comm = torch.cuda.Stream(device=device)
            self._rendered = [torch.cuda.Event() for _ in range(self.depth)]
            self._gathered = [torch.cuda.Event() for _ in range(self.depth)]
            self._in_flight = [False] * self.depth

    def buffer(self, k: int = 0) -> torch.Tensor:
        return self.send[k % self.depth]

    def begin_render(self, k: int = 0) -> None:
        """Call before rendering frame k into buffer(k)."""
        if self._comm is not None and self._in_flight[k % self.depth]:
            torch.cuda.current_stream().wait_event(self._gathered[k % self.depth])

    def gather(self, k: int = 0) -> Optional[torch.Tensor]:
        """Collects every rank's bands of frame k on rank 0 and places them in the frame.  Returns the frame on
        rank 0 (on a GPU: once the work queued here has run; see finish())."""
        src = self.buffer(k)
        if not self.collective:
            self.frame[: self.slots].copy_(src)
            return self.frame
        if self._comm is None:
            self._gather_and_place(src)
        else:
            i = k % self.depth
            self._rendered[i].record(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(self._rendered[i])
                self._gather_and_place(src)
                self._gathered[i].record(self._comm)
            self._in_flight[i] = True
        return self.frame if self.rank == 0 else None

    def _gather_and_place(self, src: torch.Tensor) -> None:
        if src.is_cuda and dist.get_backend() != "nccl":
            # gloo is a CPU backend: handed device tensors it reads device memory from host threads, with no ordering
            # against the stream that wrote it.  Bands therefore travel through pinned host memory: wait for the
            # kernels queued so far, copy out, gather host tensors, copy the gathered bands back in.
            if self._host_send is None:
                self._host_send = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                if self.rank == 0:
                    self._host_recv_all = torch.empty(self._recv_all.shape, dtype=src.dtype, pin_memory=True)
                    self._host_recv = [self._host_recv_all[r] for r in range(self.world_size)]
            self._host_send.copy_(src, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist.gather(self._host_send, self._host_recv, dst=0)
            if self.rank != 0:
                return
            self._recv_all.copy_(self._host_recv_all, non_blocking=True)
            self._place()
            torch.cuda.current_stream().synchronize()  # the pinned buffers are reused by the next call
            return
        dist.gather(src, self._recv, dst=0)
        if self.rank == 0:
            self._place()

    def _place(self) -> None:
        """recv_all[rank, slot] -> frame band slot * world_size + rank, for every rank at once (bands past the end of
        the frame land in its padding)."""
        self._frame_padded.view(self.slots, self.world_size, self.band_vals).copy_(self._recv_all.permute(1, 0, 2))

    def finish(self) -> None:
        """Makes the current stream wait for every gather queued so far."""
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)

    def frame_rows(self) -> torch.Tensor:
        """The assembled frame as [height, width, channels] (rank 0)."""
        return self.frame.view(self.n_bands * self.band_rows, self.width, self.channels)[: self.height]
