"""Tile sharding of SamplerIntegrator::Render across the GPUs of one node.

The reference renders 16x16 tiles from a shared-memory worker pool
(core/integrator.cpp:237-244, core/parallel.cpp:247-299) and merges each FilmTile under
a mutex (core/film.cpp:118-132).  Here rank r of `world` renders tiles r, r+world, ...
of the same row-major tile grid into its own film buffer (xyz + filterWeightSum per
pixel, zeros for pixels of tiles it does not own) and the buffers are merged on rank 0.

On GPUs the merge is `Comm.film_gather` (hprt_film_gather, csrc/capi_gather.hip): one
ncclReduce of the films over xGMI plus a grouped send/recv of the few box-filter
contributions that cross a tile border (core/film.h:136-143), which the root adds per
pixel in source-tile order — the order of the single-GPU film, so the result does not
depend on the number of ranks.  `gather_film` below is the same merge with
torch.distributed as the transport (gloo: CPU tests, or several ranks sharing one GPU,
which RCCL refuses): reduce(SUM) of the films, gather of the records, ordered merge on
the host (hprt_film_records_merge).
"""
import importlib

import numpy as np


def shard(rank, world):
    """Arguments for Scene.render / hprt_render: this rank's tile subset."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    return {"tile_begin": rank, "tile_end": 0, "tile_stride": world}


def gather_film(film, dist, dst=0, records=None):
    """Merge the per-rank film tensors onto `dst` (in place).  `dist` is torch.distributed.

    records: this rank's cross-tile records (`Scene.film_records()`, or an empty FILM_RECORD array) when the films were
    rendered with export_foreign=True; None when each rank merged its cross-tile contributions into its own film
    (then the plain sum is the merged film, exact as long as no pixel receives contributions from more than two ranks).
    With the RCCL backend ("nccl") a device tensor is reduced directly over xGMI; with gloo it takes a round trip
    through host memory."""
    # Agreement before the first data collective, as hprt_film_gather has it (csrc/capi_gather.hip): a rank whose own
    # arguments are unusable must not leave its peers waiting in the reduce, so every rank publishes what it is about to
    # merge and all of them raise together when the descriptions differ or one rank reports a failure.
    hprt = importlib.import_module(__package__)
    problem = None
    try:
        if not hasattr(film, "shape") or not hasattr(film, "is_cuda"):
            problem = "film is not a tensor"
        elif records is not None and np.asarray(records).dtype != hprt.FILM_RECORD:
            problem = "records are not FILM_RECORD entries"
    except Exception as e:      # (anything odd about the arguments is a local failure, reported like the others)
        problem = repr(e)
    mine = (problem, tuple(getattr(film, "shape", ())), str(getattr(film, "dtype", "")), records is not None)
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    failed = [(r, m[0]) for r, m in enumerate(everyone) if m[0] is not None]
    if failed:
        raise RuntimeError("gather_film: rank %d reported: %s; nothing was merged" % failed[0])
    if any(m[1:] != everyone[0][1:] for m in everyone):
        raise RuntimeError("gather_film: the ranks disagree about the film (shape, dtype, records): %r; nothing was merged" % ([m[1:] for m in everyone],))
    via_host = film.is_cuda and dist.get_backend() == "gloo"
    host = film.cpu() if via_host else film
    dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
    if records is not None:
        world = dist.get_world_size()
        parts = [None] * world if dist.get_rank() == dst else None
        dist.gather_object(np.ascontiguousarray(records), parts, dst=dst)
        if dist.get_rank() == dst:
            allrec = np.concatenate([np.asarray(p, hprt.FILM_RECORD) for p in parts]) if parts else np.zeros(0, hprt.FILM_RECORD)
            merged = host.cpu().numpy() if host.is_cuda else host.numpy()
            hprt.film_records_merge(merged, allrec)
            if host.is_cuda:
                import torch
                host.copy_(torch.from_numpy(merged))
    if via_host:
        film.copy_(host)
    return film


def gather_pixel_stats(stats, dist, dst=0):
    """Sum per-rank Pixel::stats tensors (int64 [H, W, 7] from Scene.pixel_stats(); zeros outside a rank's
    tiles) onto `dst`: Film::MergeFilmTile's `mergePixel.stats += tilePixel.stats` (core/film.cpp:130)."""
    return gather_film(stats, dist, dst)
