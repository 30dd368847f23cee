"""Row-band multi-GPU split of one frame: one process per GPU,
rank r renders rows band_rows(H, r, N) of the SAME frame (global y in the ray
generation, /root/reference/kernel.cu:1624-1625), then ONE gather of the packed
bands to rank 0 reassembles the image. No other collective is needed: pixels are
independent (one store per thread, kernel.cu:1682/1688).

Backend: "nccl" (= RCCL over xGMI) on GPUs; the same code runs on "gloo" with
CPU tensors, which is how the host logic is tested without GPUs."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import band_rows, interleaved_rows


def max_band_rows(height: int, world: int) -> int:
    return max(band_rows(height, r, world)[1] - band_rows(height, r, world)[0] for r in range(world))


def alloc_band(height: int, width: int, world: int, device, dtype=torch.int32) -> torch.Tensor:
    """Band buffer padded to the largest band so that every rank contributes the
    same element count to the gather (bands differ by at most one row)."""
    return torch.zeros((max_band_rows(height, world), width), dtype=dtype, device=device)


def gather_bands(band: torch.Tensor, dst: int = 0, group=None, out=None):
    """One gather of equal-sized (padded) bands to `dst`. Returns the list of
    per-rank bands on `dst`, None elsewhere. `out` lets the caller reuse buffers."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        if out is None:
            out = [torch.empty_like(band) for _ in range(world)]
        dist.gather(band, out, dst=dst, group=group)
        return out
    dist.gather(band, None, dst=dst, group=group)
    return None


def assemble_frame(gathered, height: int) -> torch.Tensor:
    """Drop each band's padding rows and stack the bands in rank order."""
    world = len(gathered)
    parts = []
    for r, g in enumerate(gathered):
        y0, y1 = band_rows(height, r, world)
        parts.append(g[: y1 - y0])
    return torch.cat(parts, dim=0)


# --- balanced variant: row blocks dealt round-robin (rt_launch_opts.interleave_*) ---
def max_interleaved_rows(height: int, world: int, block: int = 16) -> int:
    return max(len(interleaved_rows(height, r, world, block)) for r in range(world))


def alloc_interleaved(height: int, width: int, world: int, device, block: int = 16, dtype=torch.int32):
    return torch.zeros((max_interleaved_rows(height, world, block), width), dtype=dtype, device=device)


def assemble_interleaved(gathered, height: int, block: int = 16, out=None) -> torch.Tensor:
    """Scatter each rank's compact rows back to their global rows (one indexed
    copy per rank on the root GPU)."""
    world = len(gathered)
    width = gathered[0].shape[1]
    if out is None:
        out = torch.empty((height, width), dtype=gathered[0].dtype, device=gathered[0].device)
    for r, g in enumerate(gathered):
        idx = torch.as_tensor(interleaved_rows(height, r, world, block), dtype=torch.int64, device=g.device)
        out.index_copy_(0, idx, g[: idx.numel()])
    return out


def rgb24_row_words(width: int) -> int:
    """int32 words of one row in the 3-bytes-per-pixel form (`rt_launch_opts.packed24`)."""
    if width % 4:
        raise ValueError("the 24-bit row form needs a width that is a multiple of 4")
    return width * 3 // 4


def unpack_rgb24(rows24: torch.Tensor, width: int, out=None) -> torch.Tensor:
    """[rows, 3*width/4] int32 (bytes B,G,R per pixel) -> [rows, width] int32 words 0x00RRGGBB,
    the form `setPixelBuff` consumes. `out`, if given, must have its top bytes already zero."""
    rows = rows24.shape[0]
    if out is None:
        out = torch.zeros((rows, width), dtype=torch.int32, device=rows24.device)
    out.view(torch.uint8).view(rows, width, 4)[..., :3].copy_(rows24.view(torch.uint8).view(rows, width, 3))
    return out


class InterleavedGather:
    """Root-side buffers for the interleaved split: one contiguous receive buffer
    whose per-rank slices are the gather list, and a row permutation so that the
    whole frame is put in place by ONE index_select kernel per frame. With `rgb24`
    the ranks send 3 bytes per pixel (the packed word without its zero byte, a quarter
    less over xGMI) and the root widens the assembled frame back to 32-bit words."""

    def __init__(self, height: int, width: int, world: int, device, block: int = 16, dtype=torch.int32,
                 rgb24: bool = False):
        self.height, self.width, self.world, self.block, self.rgb24 = height, width, world, block, rgb24
        self.max_rows = max_interleaved_rows(height, world, block)
        self.row_words = rgb24_row_words(width) if rgb24 else width
        self.recv = torch.empty((world, self.max_rows, self.row_words), dtype=dtype, device=device)
        self.views = [self.recv[r] for r in range(world)]
        src = torch.empty(height, dtype=torch.int64)
        for r in range(world):
            rows = interleaved_rows(height, r, world, block)
            src[torch.as_tensor(rows, dtype=torch.int64)] = r * self.max_rows + torch.arange(len(rows))
        self.src_rows = src.to(device)            # frame row y comes from recv.view(-1, W)[src_rows[y]]
        self.frame = torch.zeros((height, width), dtype=dtype, device=device)
        self.frame24 = torch.empty((height, self.row_words), dtype=dtype, device=device) if rgb24 else None

    def assemble(self) -> torch.Tensor:
        if self.rgb24 and self.recv.is_cuda and self.block == 16:
            # one kernel of the library: rows home and 24 -> 32 bits (rt_multi.hip, rt_scatter_rows24)
            from . import load_library, _check
            _check(load_library().rt_assemble_rows24(self.recv.data_ptr(), self.frame.data_ptr(), self.width, self.height,
                                                     self.world, self.max_rows, torch.cuda.current_stream().cuda_stream),
                   "rt_assemble_rows24")
            return self.frame
        if self.rgb24:
            torch.index_select(self.recv.view(self.world * self.max_rows, -1), 0, self.src_rows, out=self.frame24)
            return unpack_rgb24(self.frame24, self.width, out=self.frame)
        torch.index_select(self.recv.view(self.world * self.max_rows, -1), 0, self.src_rows, out=self.frame)
        return self.frame
