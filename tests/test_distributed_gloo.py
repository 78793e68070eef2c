"""world_size-2 test of the tile-sharded Render + film gather on CPU (gloo).

Each rank produces the film of ITS tiles (tiles r, r+2, ... of the 16x16 grid) and the
films are summed onto rank 0 with thesis-pbrt-v3_amd/tiles.py::gather_film — the same
helper bench.py uses over RCCL.  On this GPU-less box the per-rank film comes from the
oracle's tile-subset render; the test proves the sharding + reduce reproduces the
single-process film BIT FOR BIT, including pixels that receive box-filter samples from
a neighbouring tile owned by the other rank."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import KILLEROO, ROOT

CROP = (0.40, 0.40 + 80 / 700.0, 0.45, 0.45 + 64 / 700.0)
SPP = 4


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import orc
    tiles = importlib.import_module("thesis-pbrt-v3_amd.tiles")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = orc.OracleScene(KILLEROO)
    o.set_film(crop=CROP, spp=SPP)
    sh = tiles.shard(rank, world)
    orc.lib.orc_render_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    orc.lib.orc_render_tiles(o._h, SPP, 2, sh["tile_begin"], sh["tile_stride"], None)
    x0, y0, x1, y1 = o.film_bounds()
    film = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
    orc.lib.orc_film_raw(o._h, film.ctypes.data_as(C.c_void_p))
    t = torch.from_numpy(film)
    own = int((film[..., 3] > 0).sum())
    tiles.gather_film(t, dist, dst=0)
    counts = torch.tensor([own], dtype=torch.int64)
    dist.all_reduce(counts)
    # the per-pixel traversal statistics shard and merge the same way (Film::MergeFilmTile, core/film.cpp:130)
    st = torch.from_numpy(o.pixel_stats().astype(np.int64))
    tiles.gather_pixel_stats(st, dist, dst=0)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel(), np.array([float(counts.item())], np.float32)]))
        np.save(out_path + ".stats.npy", st.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_reproduces_the_film(tmp_path, killeroo_oracle):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    killeroo_oracle.set_film(crop=CROP, spp=SPP)
    _, film, _, _, _ = killeroo_oracle.render(spp=SPP, threads=4)
    stats = killeroo_oracle.pixel_stats()
    killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    assert np.array_equal(got[:-1].view(np.uint32), film.ravel().view(np.uint32))
    assert np.array_equal(np.load(out + ".stats.npy"), stats.astype(np.int64)) and stats[..., 1].sum() > 0
    # some pixels were written by both ranks (cross-tile filter footprint) -> the reduce really summed
    assert got[-1] >= film.shape[0] * film.shape[1]


def test_shard_covers_every_tile_once():
    tiles = importlib.import_module("thesis-pbrt-v3_amd.tiles")
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            sh = tiles.shard(r, world)
            seen += list(range(sh["tile_begin"], 1936, sh["tile_stride"]))
        assert sorted(seen) == list(range(1936))
