"""Tile sharding of SamplerIntegrator::Render across the GPUs of one node.

The reference renders 16x16 tiles from a shared-memory worker pool
(core/integrator.cpp:237-244, core/parallel.cpp:247-299) and merges each FilmTile under
a mutex (core/film.cpp:118-132).  Here rank r of `world` renders tiles r, r+world, ...
of the same row-major tile grid into its own film buffer (xyz + filterWeightSum per
pixel, zeros for pixels of tiles it does not own) and the buffers are SUMMED onto rank 0
with one reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).  Every pixel has one
non-zero addend except the few that receive a box-filter sample from a neighbouring
tile (core/film.h:136-143), where the sum is exactly MergeFilmTile's `xyz += ...`.
"""


def shard(rank, world):
    """Arguments for Scene.render / hprt_render: this rank's tile subset."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    return {"tile_begin": rank, "tile_end": 0, "tile_stride": world}


def gather_film(film, dist, dst=0):
    """Sum the per-rank film tensors onto `dst` (in place).  `dist` is torch.distributed.

    With the RCCL backend ("nccl") the device tensor is reduced directly over xGMI.  With
    gloo (CPU rehearsals, or several ranks sharing one GPU in the tests) a device tensor
    takes a round trip through host memory."""
    if film.is_cuda and dist.get_backend() == "gloo":
        host = film.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        film.copy_(host)
    else:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def gather_pixel_stats(stats, dist, dst=0):
    """Sum per-rank Pixel::stats tensors (int64 [H, W, 7] from Scene.pixel_stats(); zeros outside a rank's
    tiles) onto `dst`: Film::MergeFilmTile's `mergePixel.stats += tilePixel.stats` (core/film.cpp:130)."""
    return gather_film(stats, dist, dst)
