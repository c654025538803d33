"""One frame over several GPUs of a node: row-band sharding and the gather of finished bands to rank 0.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" for CPU rehearsal).  The scene is
small and read-only, so every rank holds a copy; pixels are independent, so the only exchange is the gather of the
rendered rows.  The frame is cut into bands of `band_rows` rows and band b belongs to rank b % world_size: band cost
varies by an order of magnitude between sky and glass, and a fine round-robin interleave evens that out without any
run-time coordination between the processes.

The gather is `torch.distributed.gather`, which on the NCCL/RCCL backend is a group of point-to-point send/recv:
every peer's bands travel over that peer's own xGMI link into rank 0.
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

    `mine` is this rank's send buffer ([bands_per_rank, band_rows*width*channels]); rank 0 also owns `frame`
    ([n_bands, band_rows*width*channels], i.e. the frame padded to whole bands; `frame_rows()` views the real rows).
    """

    def __init__(self, width: int, height: int, rank: int, world_size: int, device: torch.device,
                 band_rows: int = DEFAULT_BAND_ROWS, channels: int = 3, dtype: torch.dtype = torch.float64):
        self.width, self.height, self.rank, self.world_size = width, height, rank, world_size
        self.band_rows, self.channels = band_rows, channels
        self.band_vals = band_rows * width * channels
        self.n_bands = n_bands(height, band_rows)
        self.slots = bands_per_rank(world_size, height, band_rows)
        self.mine = torch.zeros(self.slots, self.band_vals, dtype=dtype, device=device)
        self.frame: Optional[torch.Tensor] = None
        self._recv: Optional[List[torch.Tensor]] = None
        if rank == 0:
            self.frame = torch.zeros(self.n_bands, self.band_vals, dtype=dtype, device=device)
            if world_size > 1:
                self._recv = [torch.empty_like(self.mine) for _ in range(world_size)]

    def gather(self) -> Optional[torch.Tensor]:
        """Collects every rank's bands on rank 0 and places them in the frame.  Returns the frame on rank 0."""
        if self.world_size == 1:
            self.frame[: self.slots].copy_(self.mine)
            return self.frame
        dist.gather(self.mine, self._recv, dst=0)
        if self.rank != 0:
            return None
        for r in range(self.world_size):
            k = len(range(r, self.n_bands, self.world_size))
            self.frame[r::self.world_size] = self._recv[r][:k]
        return self.frame

    def frame_rows(self) -> torch.Tensor:
        """The assembled frame as [height, width, channels] (rank 0)."""
        return self.frame.view(self.n_bands * self.band_rows, self.width, self.channels)[: self.height]
