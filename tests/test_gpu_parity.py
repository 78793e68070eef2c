"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on
identical inputs.  Integer/index results and float results are compared BIT FOR BIT
(np.array_equal on the raw float32 arrays): the kernels restate the reference's IEEE
operation sequence, so the stated tolerance of BASELINE.json (per-pixel L-inf < 1e-4)
is met with zero difference against the oracle in its deterministic-math mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _camera_ray_set(oracle, n, seed):
    rng = np.random.default_rng(seed)
    px = rng.integers(0, 700, n).astype(np.int32)
    py = rng.integers(0, 700, n).astype(np.int32)
    s = rng.integers(0, 64, n).astype(np.int64)
    o, d = oracle.camera_rays(px, py, s)
    return o, d, np.full(n, np.inf, np.float32)


def _secondary_ray_set(oracle, n, seed):
    """Rays leaving surface points in random directions (what bounce/shadow rays look like)."""
    o, d, tmax = _camera_ray_set(oracle, n, seed)
    t, prim, _, _ = oracle.intersect(o, d, tmax)
    hit = prim >= 0
    rng = np.random.default_rng(seed + 1)
    p = o[hit] + d[hit] * t[hit, None] * np.float32(0.999)
    v = rng.normal(size=p.shape).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return p.astype(np.float32), v.astype(np.float32), np.full(p.shape[0], np.inf, np.float32)


def test_closest_hit_camera_rays(killeroo_scene, killeroo_oracle):
    o, d, tmax = _camera_ray_set(killeroo_oracle, 200000, 1)
    t0, p0, b0, c0 = killeroo_oracle.intersect(o, d, tmax)
    t1, p1, b1, c1 = killeroo_scene.intersect(o, d, tmax, count=True)
    assert np.array_equal(p0, p1), "primitive index mismatch on %d rays" % int((p0 != p1).sum())
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32))
    # work counters: nodes fetched / entered, triangle tests, sphere tests
    assert [int(x) for x in c1] == [c0["nodes_fetched"], c0["nodes_entered"], c0["tri_tests"], c0["sphere_tests"]]


def test_closest_hit_secondary_rays(killeroo_scene, killeroo_oracle):
    o, d, tmax = _secondary_ray_set(killeroo_oracle, 200000, 2)
    t0, p0, b0, c0 = killeroo_oracle.intersect(o, d, tmax)
    t1, p1, b1, c1 = killeroo_scene.intersect(o, d, tmax, count=True)
    assert np.array_equal(p0, p1)
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32))
    assert [int(x) for x in c1] == [c0["nodes_fetched"], c0["nodes_entered"], c0["tri_tests"], c0["sphere_tests"]]


def test_any_hit_shadow_rays(killeroo_scene, killeroo_oracle):
    # unnormalised segment rays towards the light, tMax = 1 - ShadowEpsilon (core/interaction.h:73-78)
    o, d, _ = _secondary_ray_set(killeroo_oracle, 200000, 3)
    # end points just outside the emitter sphere (centre (150,120,20), radius 3), on the side facing the origin
    light = np.array([150.0, 120.0, 20.0], np.float32)
    to_o = o - light[None, :]
    target = light[None, :] + to_o / np.linalg.norm(to_o, axis=1, keepdims=True) * np.float32(3.01)
    seg = (target - o).astype(np.float32)
    tmax = np.full(o.shape[0], np.float32(1) - np.float32(0.0001), np.float32)
    occ0, c0 = killeroo_oracle.occluded(o, seg, tmax)
    occ1, c1 = killeroo_scene.occluded(o, seg, tmax, count=True)
    assert np.array_equal(occ0, occ1)
    assert [int(x) for x in c1] == [c0["nodes_fetched_p"], c0["nodes_entered_p"], c0["tri_tests_p"], c0["sphere_tests_p"]]
    assert 0 < occ0.mean() < 1


def test_edge_cases_empty_and_degenerate_rays(killeroo_scene, killeroo_oracle):
    e = np.zeros((0, 3), np.float32)
    t, p, b = killeroo_scene.intersect(e, e, np.zeros(0, np.float32))
    assert t.shape == (0,) and p.shape == (0,)
    # axis-aligned directions (1/0 = inf in invDir), zero tMax, rays starting outside the scene bounds
    o = np.array([[0, 0, 500], [0, 0, 500], [5000, 0, 0], [0, 63, -110], [400, 20, 30]], np.float32)
    d = np.array([[0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [-1, 0, 0]], np.float32)
    tmax = np.array([np.inf, np.inf, np.inf, 0.0, 1e-3], np.float32)
    t0, p0, b0, _ = killeroo_oracle.intersect(o, d, tmax)
    t1, p1, b1 = killeroo_scene.intersect(o, d, tmax)
    assert np.array_equal(p0, p1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))


def test_per_sample_radiance(killeroo_scene, killeroo_oracle):
    rng = np.random.default_rng(7)
    n = 60000
    px = rng.integers(0, 700, n).astype(np.int32)
    py = rng.integers(0, 700, n).astype(np.int32)
    s = rng.integers(0, 256, n).astype(np.int64)
    L0 = killeroo_oracle.sample_radiance(px, py, s)
    L1 = killeroo_scene.sample_radiance(px, py, s)
    bad = np.any(L0.view(np.uint32) != L1.view(np.uint32), axis=1)
    assert not bad.any(), "%d of %d samples differ; max |d| = %g" % (int(bad.sum()), n, float(np.abs(L0 - L1).max()))
    assert L0.max() > 0


def test_render_crop_film_and_counters(hprt, killeroo_model, killeroo_scene, killeroo_oracle):
    # 96x80 crop window of the 700x700 frame that contains killeroo, floor and the light's highlight
    opt = killeroo_model.options.copy()
    crop = (0.40, 0.40 + 96 / 700.0, 0.45, 0.45 + 80 / 700.0)
    for i in range(4):
        opt.crop[i] = crop[i]
    opt.spp = 16
    killeroo_oracle.set_film(crop=crop, spp=16)
    rgb0, film0, c0, _, _ = killeroo_oracle.render(spp=16, threads=8)
    film1, st = killeroo_scene.render(opt, count_work=True)
    assert film1.shape == film0.shape
    assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32)), \
        "film differs in %d pixels, max |d| %g" % (int(np.any(film0 != film1, axis=2).sum()), float(np.abs(film0 - film1).max()))
    rgb1 = hprt.film_resolve(film1, opt.film_scale)
    assert np.array_equal(rgb0.view(np.uint32), rgb1.view(np.uint32))
    # scratch memory filled with garbage first (large floats / small indices): the film may not depend on it
    try:
        for byte in (0x7F, 0x01):
            killeroo_scene.debug_poison(byte)
            assert np.array_equal(killeroo_scene.render(opt)[0].view(np.uint32), film1.view(np.uint32)), hex(byte)
    finally:
        killeroo_scene.debug_poison(None)
    assert st["camera_rays"] == c0["camera_rays"] and st["rays"] == c0["rays"] and st["shadow_rays"] == c0["shadow_rays"]
    assert st["nodes_fetched"] == c0["nodes_fetched"] and st["nodes_fetched_p"] == c0["nodes_fetched_p"]
    assert st["nodes_entered"] == c0["nodes_entered"] and st["nodes_entered_p"] == c0["nodes_entered_p"]
    assert st["tri_tests"] == c0["tri_tests"] and st["tri_tests_p"] == c0["tri_tests_p"]
    killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)


def test_full_size_frame_is_independent_of_batching_and_sharding(killeroo_model, killeroo_scene):
    """BASELINE.json's config[1] at its full size (700x700, 256 spp, 125 M paths, 737 M rays) is too large for the oracle
    in test time, so the whole frame is held to size-independent properties: the film and every work counter must not
    depend on how the samples are cut into wavefront batches (one batch of 125 M paths, or batches of 100 and of 37
    samples per pixel) nor on how the tiles are cut into shards (three interleaved shards summed, as MergeFilmTile
    does); the sampled per-sample parity against the oracle is test_per_sample_radiance."""
    opt = killeroo_model.options.copy()
    opt.spp = 256                      # the baked fixture carries the 8 spp of the reference's regression image
    assert (opt.xres, opt.yres) == (700, 700)
    film_a, st_a = killeroo_scene.render(opt, count_work=True)                       # automatic: one batch
    film_b, st_b = killeroo_scene.render(opt, count_work=True, spp_chunk=100)         # 100 + 100 + 56
    film_c, st_c = killeroo_scene.render(opt, count_work=True, spp_chunk=37)
    keys = ("camera_rays", "rays", "shadow_rays", "nodes_fetched", "nodes_fetched_p", "nodes_entered", "nodes_entered_p",
            "tri_tests", "tri_tests_p", "sphere_tests", "sphere_tests_p")
    assert st_a["camera_rays"] == 700 * 700 * 256 and st_a["rays"] + st_a["shadow_rays"] > 7 * 10 ** 8
    for st in (st_b, st_c):
        assert [st[k] for k in keys] == [st_a[k] for k in keys]
    assert np.array_equal(film_a.view(np.uint32), film_b.view(np.uint32)) and np.array_equal(film_a.view(np.uint32), film_c.view(np.uint32))
    # filterWeightSum (box filter, every weight 1): 256 except where a sample with an exactly-zero or rounded-up offset
    # also lands in a neighbour
    w = film_a[..., 3]
    assert np.isfinite(film_a).all() and (np.abs(w - 256) <= 2).all() and (w == 256).mean() > 0.97, (float(w.min()), float(w.max()), float((w == 256).mean()))
    total = np.zeros_like(film_a)
    for r in range(3):
        part, _ = killeroo_scene.render(opt, tile_begin=r, tile_stride=3)
        total += part
    assert np.array_equal(total.view(np.uint32), film_a.view(np.uint32))


def _rank_worker(rank, world, port, out_path, crop, spp):
    import importlib, os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    from conftest import KILLEROO, ROOT
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    hprt = importlib.import_module("thesis-pbrt-v3_amd")
    tiles = importlib.import_module("thesis-pbrt-v3_amd.tiles")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = hprt.Model.load(KILLEROO); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh, device=0)
    opt = model.options.copy()
    for i in range(4):
        opt.crop[i] = crop[i]
    opt.spp = spp
    x0, y0, x1, y1 = opt.film_bounds()
    film = torch.zeros((y1 - y0, x1 - x0, 4), dtype=torch.float32, device="cuda:0")
    scene.render(opt, film_ptr=film.data_ptr(), export_foreign=True, **tiles.shard(rank, world))
    rec = scene.film_records()
    tiles.gather_film(film, dist, dst=0, records=rec)
    n = torch.tensor([len(rec)], dtype=torch.int64)
    dist.all_reduce(n)
    if rank == 0:
        np.save(out_path, film.cpu().numpy())
        np.save(out_path + ".nrec.npy", n.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_sharing_the_gpu_reproduce_the_film(tmp_path, killeroo_oracle, killeroo_scene, killeroo_model, world):
    """N > 1 path on the 1-GPU box: `world` processes render interleaved tiles of the WHOLE 700x700 frame on cuda:0 with
    HPRT_RENDER_EXPORT_FOREIGN; films are summed and the cross-tile records merged in source-tile order (tiles.gather_film:
    the gloo twin of hprt_film_gather, RCCL refuses ranks that share a device).  The merged film must equal the oracle's
    single-process film AND the product's own unsharded film bit for bit — including tile-corner pixels that receive
    samples from tiles of several other ranks (world 3: the right, lower and lower-right neighbours of a tile belong to
    three different ranks)."""
    import socket
    import torch.multiprocessing as mp
    crop = (0.0, 1.0, 0.0, 1.0)
    spp = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "film.npy")
    mp.spawn(_rank_worker, args=(world, port, out, crop, spp), nprocs=world, join=True)
    got = np.load(out)
    assert int(np.load(out + ".nrec.npy")[0]) > 100          # the frame does have cross-tile contributions
    opt = killeroo_model.options.copy(); opt.spp = spp
    plain, _ = killeroo_scene.render(opt)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32))
    killeroo_oracle.set_film(crop=crop, spp=spp)
    _, film0, _, _, _ = killeroo_oracle.render(spp=spp, threads=16)
    killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    assert np.array_equal(got.view(np.uint32), film0.view(np.uint32))


def test_rccl_film_gather_through_the_c_abi(hprt, killeroo_model, killeroo_scene):
    """hprt_comm_* / hprt_film_gather on real RCCL with the one rank a 1-GPU box allows: ncclCommInitRank from a unique id,
    ncclAllGather of the record counts, ncclReduce of the film, and the device-side ordered merge of the exported cross-tile
    records must turn an EXPORT_FOREIGN render into exactly the plain render's film; hprt_film_gather_local (one process,
    one scene per GPU: ncclCommInitAll) likewise.  Multi-GPU RCCL runs only in the driver's SCALE bench."""
    import torch
    opt = killeroo_model.options.copy(); opt.spp = 4
    plain, _ = killeroo_scene.render(opt)
    x0, y0, x1, y1 = opt.film_bounds()
    n_pix = (x1 - x0) * (y1 - y0)
    film = torch.zeros((y1 - y0, x1 - x0, 4), dtype=torch.float32, device="cuda:0")
    killeroo_scene.render(opt, film_ptr=film.data_ptr(), export_foreign=True)
    rec = killeroo_scene.film_records()
    assert len(rec) > 100 and not np.array_equal(film.cpu().numpy().view(np.uint32), plain.view(np.uint32))
    # the records name pixels of OTHER tiles; several records may meet in one pixel
    comm = hprt.Comm(hprt.Comm.unique_id(), 0, 1, device=0)
    assert comm.info() == {"rank": 0, "n_ranks": 1, "device": 0}
    comm.film_gather(killeroo_scene, film.data_ptr(), n_pix, root=0, stream=torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(film.cpu().numpy().view(np.uint32), plain.view(np.uint32))
    # host twin of the merge (what the gloo path uses)
    killeroo_scene.render(opt, film_ptr=film.data_ptr(), export_foreign=True)
    host = film.cpu().numpy().copy()
    hprt.film_records_merge(host, rec[::-1])
    assert np.array_equal(host.view(np.uint32), plain.view(np.uint32))
    # single-process variant, library-owned film
    killeroo_scene.render(opt, export_foreign=True)
    raw = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
    hprt._check(hprt.lib.hprt_film_read(killeroo_scene._h, hprt._ptr(raw), n_pix))
    rec2 = killeroo_scene.film_records()
    hprt.film_gather_local([killeroo_scene], None, n_pix, root=0)
    own = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
    hprt._check(hprt.lib.hprt_film_read(killeroo_scene._h, hprt._ptr(own), n_pix))
    bad = np.any(own.view(np.uint32) != plain.view(np.uint32), axis=2)
    assert not bad.any(), "local gather: %d pixels differ at %s; records equal the first render's: %s; raw + host merge equals plain: %s" % (
        int(bad.sum()), np.argwhere(bad)[:6].tolist(), np.array_equal(rec2, rec),
        np.array_equal(hprt.film_records_merge(raw.copy(), rec2).view(np.uint32), plain.view(np.uint32)))
    # a gather after a render without the flag is refused
    killeroo_scene.render(opt, film_ptr=film.data_ptr())
    with pytest.raises(hprt.HprtError):
        comm.film_gather(killeroo_scene, film.data_ptr(), n_pix)
    # ... and it was refused THROUGH the collective (the failing rank still takes part in the count exchange, so that no peer is
    # left waiting): the communicator is as usable as before — another failure of the same kind, then a valid gather
    killeroo_scene.render(opt, film_ptr=film.data_ptr(), export_foreign=True)
    other = torch.zeros_like(film)
    with pytest.raises(hprt.HprtError, match="not the buffer"):      # a film buffer the last render did not write
        comm.film_gather(killeroo_scene, other.data_ptr(), n_pix)
    with pytest.raises(hprt.HprtError):                                # a film of another size
        comm.film_gather(killeroo_scene, film.data_ptr(), n_pix - 1)
    comm.film_gather(killeroo_scene, None, n_pix, root=0, stream=torch.cuda.current_stream().cuda_stream)   # NULL: the buffer the last render wrote
    assert np.array_equal(film.cpu().numpy().view(np.uint32), plain.view(np.uint32))
    # the library-owned film is stale after a render into a caller's buffer: reading it is refused, not answered with old pixels
    with pytest.raises(hprt.HprtError):
        hprt._check(hprt.lib.hprt_film_read(killeroo_scene._h, hprt._ptr(own), n_pix))
    del comm
    hprt.lib.hprt_film_gather_local_shutdown()
    # ... after which the single-process gather simply builds its communicators again
    killeroo_scene.render(opt, export_foreign=True)
    hprt.film_gather_local([killeroo_scene], None, n_pix, root=0)
    hprt._check(hprt.lib.hprt_film_read(killeroo_scene._h, hprt._ptr(own), n_pix))
    assert np.array_equal(own.view(np.uint32), plain.view(np.uint32))
    hprt.lib.hprt_film_gather_local_shutdown()


def _rccl_rank_worker(rank, world, port, out_path, spp):
    """One process per GPU, as bench.py runs them: tile shard rendered with EXPORT_FOREIGN on device `rank`, films merged by
    hprt_film_gather over a communicator whose id travelled through a gloo store."""
    import importlib, os
    import torch
    import torch.distributed as dist
    from conftest import KILLEROO
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    hprt = importlib.import_module("thesis-pbrt-v3_amd")
    tiles = importlib.import_module("thesis-pbrt-v3_amd.tiles")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    ids = [hprt.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    model = hprt.Model.load(KILLEROO); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh, device=rank)
    opt = model.options.copy(); opt.spp = spp
    x0, y0, x1, y1 = opt.film_bounds()
    film = torch.zeros((y1 - y0, x1 - x0, 4), dtype=torch.float32, device="cuda:%d" % rank)
    comm = hprt.Comm(ids[0], rank, world, device=rank)
    info = comm.info()
    assert (info["rank"], info["n_ranks"], info["device"]) == (rank, world, rank)
    scene.render(opt, film_ptr=film.data_ptr(), export_foreign=True, **tiles.shard(rank, world))
    comm.film_gather(scene, film.data_ptr(), (x1 - x0) * (y1 - y0), root=0)
    torch.cuda.synchronize(rank)
    if rank == 0:
        np.save(out_path, film.cpu().numpy())
    dist.barrier()
    del comm
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, "all"])
def test_rccl_film_gather_one_process_per_gpu(tmp_path, killeroo_scene, killeroo_model, world):
    """hprt_film_gather between processes that own one GPU each (RCCL over xGMI): world = every GPU of the box (at most 4),
    skipped on a 1-GPU box — where the same worker still runs as a world of one, so that the code of the multi-GPU case is
    exercised wherever the suite runs.  The merged film must equal the unsharded film bit for bit."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if world == "all":
        world = min(torch.cuda.device_count(), 4)
        if world < 2:
            pytest.skip("one GPU: RCCL refuses two ranks on one device (the N > 1 path runs on gloo in test_ranks_sharing_the_gpu_*)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "film.npy")
    mp.spawn(_rccl_rank_worker, args=(world, port, out, 2), nprocs=world, join=True)
    opt = killeroo_model.options.copy(); opt.spp = 2
    plain, _ = killeroo_scene.render(opt)
    assert np.array_equal(np.load(out).view(np.uint32), plain.view(np.uint32))


def test_film_gather_local_over_every_gpu_of_the_box(hprt, killeroo_model, killeroo_scene):
    """hprt_film_gather_local — ONE process driving several GPUs, the way examples/hprt_render.cpp and a pbrt-side adapter do: one
    scene per device, tiles dealt round-robin, ncclCommInitAll + ncclReduce, records merged on the root.  Needs two GPUs (the
    one-GPU form of the call is part of test_rccl_film_gather_through_the_c_abi)."""
    import torch
    n = min(torch.cuda.device_count(), 4)
    if n < 2:
        pytest.skip("one GPU")
    bvh = hprt.Bvh(killeroo_model)
    scenes = [hprt.Scene(killeroo_model, bvh, device=g) for g in range(n)]
    opt = killeroo_model.options.copy(); opt.spp = 2
    x0, y0, x1, y1 = opt.film_bounds()
    n_pix = (x1 - x0) * (y1 - y0)
    for g, sc in enumerate(scenes):
        sc.render(opt, export_foreign=True, tile_begin=g, tile_stride=n)      # library-owned films
    hprt.film_gather_local(scenes, None, n_pix, root=0)
    merged = np.zeros((y1 - y0, x1 - x0, 4), np.float32)
    hprt._check(hprt.lib.hprt_film_read(scenes[0]._h, hprt._ptr(merged), n_pix))
    plain, _ = killeroo_scene.render(opt)
    assert np.array_equal(merged.view(np.uint32), plain.view(np.uint32))


def test_bench_launches_its_own_ranks(tmp_path, killeroo_oracle):
    """`python bench.py --gpus 2` without WORLD_SIZE starts two fresh rank processes itself (before touching the GPU) and
    reports the world size the merge actually ran over.  On the 1-GPU box the ranks share cuda:0 (--rehearse-on-one-gpu,
    gloo transport: RCCL refuses duplicate devices); everything else is the code path of the driver's multi-GPU runs:
    fixed total spp, tiles dealt round-robin, EXPORT_FOREIGN renders, merged film on rank 0 == the oracle's film."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    dump = str(tmp_path / "film.npy")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--workload", "killeroo-simple", "--spp", "4",
           "--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline", "--dump-film", dump]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] is None and line["world_size"] == 2 and line["scaling"] == "strong" and line["config"]["spp_total"] == 4
    assert line["rays_per_step"] > 0 and line["value"] > 0
    got = np.load(dump)
    killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=4)
    _, film0, c0, _, _ = killeroo_oracle.render(spp=4, threads=16)
    killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    assert np.array_equal(got.view(np.uint32), film0.view(np.uint32))
    assert line["msamples_per_s"] > 0 and abs(line["rays_per_step"] / (c0["rays"] + c0["shadow_rays"]) - 1) < 0.5
    # a rank count that does not match the flag is refused, not mislabelled
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True)
    assert r.returncode == 2 and "refusing" in r.stderr
    # a rank that dies in start-up ends the run at once: rank 0 is then waiting in the rendezvous for a peer that will never
    # come — the launcher must terminate it and return non-zero within seconds, not sit there until somebody's time limit
    import time
    t0 = time.time()
    r = subprocess.run(cmd, env=dict(env, HPRT_TEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60, (r.returncode, time.time() - t0)
    assert "[rank 1]" in r.stderr and "terminating the other ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no result line from a broken run


def test_per_pixel_statistics_match(hprt, orc, killeroo_scene, killeroo_oracle, tmp_path):
    """The fork's heat-map data (Pixel::stats / Film::WriteGeneralStats, core/film.cpp:170-264): per pixel, the sums over
    all rays of the pixel's samples of primitive tests and leaf / interior node traversals, closest-hit and any-hit —
    identical to the oracle's for every pixel, and written in the reference's text-matrix format."""
    opt = killeroo_scene._model.options.copy()
    for i, v in enumerate((0.37, 0.63, 0.41, 0.66)):
        opt.crop[i] = v
    opt.spp = 8
    killeroo_oracle.set_film(crop=(0.37, 0.63, 0.41, 0.66), spp=8)
    killeroo_oracle.render(threads=8)
    ref = killeroo_oracle.pixel_stats()
    film, st = killeroo_scene.render(opt, pixel_stats=True)
    got = killeroo_scene.pixel_stats()
    assert got.shape == ref.shape and ref[..., 1].sum() > 0
    assert np.array_equal(got, ref)
    assert int(got[..., 1].sum()) == st["tri_tests"] + st["sphere_tests"] and int(got[..., 0].sum()) == st["camera_rays"]
    # tile-sharded: the two halves add up (what the multi-GPU gather does with the film)
    killeroo_scene.render(opt, tile_begin=0, tile_stride=2, pixel_stats=True); a = killeroo_scene.pixel_stats()
    killeroo_scene.render(opt, tile_begin=1, tile_stride=2, pixel_stats=True); b = killeroo_scene.pixel_stats()
    assert np.array_equal(a + b, ref)
    prefix = str(tmp_path / "killeroo")
    hprt.write_pixel_stats(prefix, got)
    m = np.loadtxt(prefix + "-leafNodeTraversalsP.txt", dtype=np.uint64)
    assert np.array_equal(m, ref[..., 4])
    assert np.loadtxt(prefix + "-kdTreeNodeTraversals.txt").sum() == 0
    with pytest.raises(hprt.HprtError):
        killeroo_scene.render(opt); killeroo_scene.pixel_stats()


@pytest.mark.parametrize("name", ["dodecahedron", "killeroo", "simple_instanced"])
def test_reference_regression_scenes_full_frame(hprt, orc, name):
    """The reference's other regression scenes (tests/golden/make_fixtures.py), whole 700x700 frame at their 8 spp:
    the film state of the HIP path equals the oracle's bit for bit — and the oracle reproduces the reference's
    checked-in renders of exactly these scenes (tests/test_oracle_pins.py)."""
    import os
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, name + ".hprt")
    model = hprt.Model.load(path); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh)
    oracle = orc.OracleScene(path)
    _, film0, c0, _, _ = oracle.render(threads=16)
    film1, st = scene.render(count_work=True)
    assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32))
    film_plain, st_plain = scene.render()      # plain render: dead rays not traced, same film
    assert np.array_equal(film0.view(np.uint32), film_plain.view(np.uint32)) and st_plain["rays"] <= st["rays"]
    assert st["rays"] == c0["rays"] and st["shadow_rays"] == c0["shadow_rays"] and st["nodes_fetched"] == c0["nodes_fetched"]
    assert st["tri_tests"] == c0["tri_tests"] and st["sphere_tests_p"] == c0["sphere_tests_p"]


def test_unsupported_depth_is_refused(hprt, killeroo_model, killeroo_scene):
    """maxdepth beyond the reference's 1,000 sampler dimensions (5 + 8 per bounce) is refused, not sampled from nowhere."""
    opt = killeroo_model.options.copy()
    opt.max_depth = 124
    with pytest.raises(hprt.HprtError) as e:
        killeroo_scene.render(opt)
    assert e.value.code == hprt.E_UNSUPPORTED


def test_untraced_light_rays_change_nothing(hprt, killeroo_model, killeroo_scene, killeroo_oracle):
    """EstimateDirect's BSDF-sampled ray (core/integrator.cpp:176-190) is traced only to learn whether its closest hit is the
    emitter.  A plain render skips it when the quadric pre-test proves that the emitter's Intersect would return false for
    it whatever tMax is (tests/test_oracle_pins.py holds the pre-test to the interval test): the film must be the film of
    the render that traces every ray, bit for bit, with far fewer closest-hit rays; a counting render traces the
    reference's full ray set unless told to count what a plain render traces."""
    opt = killeroo_model.options.copy()
    crop = (0.30, 0.30 + 160 / 700.0, 0.40, 0.40 + 128 / 700.0)
    for i in range(4):
        opt.crop[i] = crop[i]
    opt.spp = 32
    film_all, st_all = killeroo_scene.render(opt, trace_all=True)
    film_cut, st_cut = killeroo_scene.render(opt)
    assert np.array_equal(film_all.view(np.uint32), film_cut.view(np.uint32))
    assert st_cut["shadow_rays"] == st_all["shadow_rays"] and st_cut["camera_rays"] == st_all["camera_rays"]
    assert st_cut["rays"] < 0.8 * st_all["rays"]
    _, st_ref = killeroo_scene.render(opt, count_work=True)                        # the reference's ray set and counters
    _, st_traced = killeroo_scene.render(opt, count_work=True, count_traced=True)   # what the plain render traced
    assert st_ref["rays"] == st_all["rays"] and st_traced["rays"] == st_cut["rays"]
    assert st_traced["nodes_fetched"] < st_ref["nodes_fetched"] and st_traced["nodes_fetched_p"] == st_ref["nodes_fetched_p"]
    killeroo_oracle.set_film(crop=crop, spp=32)
    _, film0, c0, _, _ = killeroo_oracle.render(spp=32, threads=8)
    killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    assert np.array_equal(film0.view(np.uint32), film_cut.view(np.uint32))
    assert c0["rays"] == st_ref["rays"] and c0["nodes_fetched"] == st_ref["nodes_fetched"]


_FEW_WAVES_SCRIPT = r"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
hprt = importlib.import_module("thesis-pbrt-v3_amd")
model = hprt.Model.load(sys.argv[2]); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh, device=0)
opt = model.options.copy(); opt.spp = 1
for i, v in enumerate((0.3, 0.6, 0.3, 0.6)):
    opt.crop[i] = v
film, _ = scene.render(opt)
np.save(sys.argv[3], film)
"""


def test_irregular_sample_buffer_grows_on_demand(tmp_path, killeroo_oracle):
    """The list of camera samples whose box-filter footprint is not their own pixel is written with a guessed capacity; when the
    count exceeds it (very high spp) the pass is repeated with the counted size.  HPRT_IRREGULAR_CAP=16 makes a small render
    take that second pass; the film must be the oracle's."""
    import os, subprocess, sys
    from conftest import KILLEROO, ROOT
    out = str(tmp_path / "film.npy")
    env = dict(os.environ, HPRT_IRREGULAR_CAP="16")
    r = subprocess.run([sys.executable, "-c", _FEW_WAVES_SCRIPT.replace("opt.spp = 1", "opt.spp = 16"), ROOT, KILLEROO, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    killeroo_oracle.set_film(crop=(0.3, 0.6, 0.3, 0.6), spp=16)
    try:
        _, film0, _, _, _ = killeroo_oracle.render(spp=16, threads=16)
    finally:
        killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    assert (film0[..., 3] != 16).sum() > 16          # more irregular samples than the forced capacity
    assert np.array_equal(got.view(np.uint32), film0.view(np.uint32))


def test_few_waves_working_through_many_queue_chunks(tmp_path, killeroo_oracle):
    """The persistent traversal waves draw rays from the queue head in chunks and prefetch the next queue entries while the
    current rays load.  A wave that exhausts its chunk and then draws the ADJACENT chunk (nobody else drew in between) must
    not mistake the empty prefetch of the exhausted chunk for the first entries of the new one: that happened in round 2
    whenever waves started staggered (other processes or streams on the card) and showed as rare film mismatches.  Here one
    workgroup (HPRT_TRACE_MAX_BLOCKS=1, 64-ray chunks) traces every queue of a render, so adjacent draws are the rule."""
    import os, subprocess, sys
    from conftest import KILLEROO, ROOT
    out = str(tmp_path / "film.npy")
    env = dict(os.environ, HPRT_TRACE_MAX_BLOCKS="1", HPRT_TRACE_CHUNK_MAX="64")
    r = subprocess.run([sys.executable, "-c", _FEW_WAVES_SCRIPT, ROOT, KILLEROO, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    killeroo_oracle.set_film(crop=(0.3, 0.6, 0.3, 0.6), spp=1)
    try:
        _, film0, _, _, _ = killeroo_oracle.render(spp=1, threads=16)
    finally:
        killeroo_oracle.set_film(crop=(0, 1, 0, 1), spp=8)
    bad = (got.view(np.uint32) != film0.view(np.uint32)).any(axis=2)
    assert not bad.any(), "%d of %d pixels differ from the oracle's film" % (int(bad.sum()), bad.size)


def test_bench_line_holds_fractions_of_measured_peaks():
    """One short single-GPU run of bench.py: every figure the `roofline` object calls a fraction is one (of a peak measured or
    specified: never above 1), the two probes it runs inside the process report plausible MI355X rates, and the committed counter
    summary carries the stamp that `counters_stale` is decided by."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "killeroo-simple", "--steps", "1", "--warmup", "1", "--no-secondary", "--no-cpu-baseline",
           "--no-trace-all"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    roof = line["roofline"]
    assert line["metric"] and line["n_gpus"] == 1 and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 3000 < roof["peak_measured"]["best"] <= 8000           # a stream copy on an MI355X: 5-6.5 TB/s
    g = roof["gather"]
    assert 100 < g["peak_measured"]["best"] < 400 and 0 < g["frac"] <= 1 and g["peak_measured"]["bvh_like"]["best"] <= g["peak_measured"]["best"] * 1.02
    assert isinstance(roof["counters_stale"], bool) and roof["counters_code_object_sha256"] and roof["library_code_object_sha256"]
    # plain renders take the leaf-exact wide walk, and the line says what it fetched per ray (counted by its own profile variant)
    assert roof["kernel"] == "k_walk4<closest>" and 1 < roof["walk"]["records_per_ray"] < 100 and 0 < roof["walk"]["leaf_iterations_per_ray"] < 50
    assert abs(roof["walk"]["requests_per_ray"] - (4 * roof["walk"]["records_per_ray"] + 3 * roof["walk"]["leaf_iterations_per_ray"])) < 0.05
    if roof["frac"] is not None:
        assert 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
        assert roof["bound"] in ("hbm", "valu_issue", "l1_gather") and all(0 < v <= 1 for v in roof["bound_candidates"].values())
        for name, k in roof["per_kernel"].items():
            for f in ("hbm_frac", "valu_issue_frac", "useful_lane_frac", "lane_utilisation", "wait_frac"):
                assert f not in k or 0 <= k[f] <= 1.0001, (name, f, k[f])


def _hard_rays(nodes, n, seed):
    """Rays that exercise the corners of the bounds test: zero direction components of either sign (1 / 0 = inf, 0 * inf = NaN),
    origins exactly on planes of BVH nodes, finite segments, rays aimed at leaves from far outside."""
    lo = nodes[:, 0:3].view(np.float32); hi = nodes[:, 3:6].view(np.float32)
    rng = np.random.default_rng(seed)
    ext = hi[0] - lo[0]
    o = (lo[0] + rng.uniform(-0.2, 1.2, (n, 3)) * ext).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 8
    d[:k, 0] = 0.0; d[k:2 * k, 1] = -0.0; d[2 * k:3 * k, 0] = 0.0; d[2 * k:3 * k, 2] = -0.0
    pick = rng.integers(0, nodes.shape[0], 2 * k); ax = rng.integers(0, 3, 2 * k)
    o[np.arange(2 * k), ax] = np.where(rng.integers(0, 2, 2 * k) == 1, hi[pick, ax], lo[pick, ax])
    aim = rng.integers(0, nodes.shape[0], k)
    d[3 * k:4 * k] = ((lo[aim] + hi[aim]) * np.float32(0.5) - o[3 * k:4 * k]).astype(np.float32)
    tmax = np.full(n, np.inf, np.float32)
    tmax[n // 2:] = rng.uniform(0, 2.0, n - n // 2).astype(np.float32) * np.float32(np.linalg.norm(ext) / max(1e-6, float(np.abs(d).mean())))
    return o, d, tmax


def _far_small_scene_text():
    """2,000 small triangles (0.01-0.5 units) in a unit cube 1e5 units from the origin: the wide records' grid steps are below the float
    spacing of the coordinates there, so the outward rounding of the quantised boxes is what keeps every leaf reachable."""
    rng = np.random.default_rng(23)
    c = rng.uniform(0, 1, (2000, 1, 3)); e = rng.normal(size=(2000, 3, 3)) * rng.uniform(0.01, 0.5, (2000, 1, 1))
    P = (c + e + np.array([1e5, -3e4, 7e4])).astype(np.float32).reshape(-1, 3)
    pts = " ".join("%.9g" % v for v in P.ravel()); idx = " ".join(str(i) for i in range(P.shape[0]))
    return ('LookAt 100003 -30000 70000  100000 -30000 70000  0 0 1\nCamera "perspective" "float fov" [40]\nSampler "halton" "integer pixelsamples" [2]\n'
            'Integrator "path" "integer maxdepth" [3]\nFilm "image" "integer xresolution" [64] "integer yresolution" [64] "string filename" ["x.pfm"]\n'
            'Accelerator "bvh"\nWorldBegin\nLightSource "point" "point from" [100002 -29999 70002] "color I" [30 30 30]\nMaterial "matte" "color Kd" [.5 .5 .5]\n'
            'Shape "trianglemesh" "integer indices" [%s] "point P" [%s]\nWorldEnd\n' % (idx, pts))


@pytest.mark.parametrize("fixture", ["killeroo.hprt", "living_room.hprt", "killeroo_simple.hprt", "simple_instanced.hprt", "far_small"])
def test_wide_and_binary_walks_agree(hprt, orc, fixture):
    """Plain calls take the leaf-exact four-wide walk (k_walk4, csrc/wide_bvh.h); hprt_debug_wide_walk(0) keeps the binary walk
    (k_trace).  Both must return the oracle's hits bit for bit — also for rays on which the bounds test meets inf and NaN."""
    import ctypes as C
    import os
    from conftest import GOLDEN
    import tempfile
    if fixture == "far_small":
        d_ = tempfile.mkdtemp()
        open(os.path.join(d_, "far.pbrt"), "w").write(_far_small_scene_text())
        model = hprt.Model.parse(os.path.join(d_, "far.pbrt"))
        path = os.path.join(d_, "far.hprt"); model.save(path)
    else:
        path = os.path.join(GOLDEN, fixture)
        model = hprt.Model.load(path)
    if fixture == "living_room.hprt":      # the oracle reads finished pyramids only
        path = os.path.join(tempfile.mkdtemp(), "expanded.hprt"); model.save(path)
    bvh = hprt.Bvh(model)
    scene = hprt.Scene(model, bvh)
    oracle = orc.OracleScene(path)
    nodes, _ = bvh.arrays()
    n = 400000 if fixture == "killeroo.hprt" else 150000
    if fixture == "far_small":      # (and the frame: plain render = wide walk, counting render = binary walk, both against the oracle)
        oracle.set_film(spp=2)
        _, film0, c0, _, _ = oracle.render(spp=2, threads=4)
        film1, _ = scene.render(model.options)
        film2, st2 = scene.render(model.options, count_work=True)
        assert np.array_equal(film0.view(np.uint32), film1.view(np.uint32)) and np.array_equal(film0.view(np.uint32), film2.view(np.uint32))
        assert st2["nodes_fetched"] == c0["nodes_fetched"] and float(film0[..., :3].max()) > 0
    o, d, tmax = _hard_rays(nodes, n, 17)
    instanced = fixture == "simple_instanced.hprt"      # (spheres in an object instance: the two-level walk and the quadric tests)
    if instanced:
        t0, p0, i0, b0, _ = oracle.intersect_inst(o, d, tmax)
    else:
        t0, p0, b0, _ = oracle.intersect(o, d, tmax)
    occ0, _ = oracle.occluded(o, d, tmax)
    hprt.lib.hprt_debug_wide_walk.argtypes = [C.c_int]
    try:
        for wide in (1, 0):
            hprt.lib.hprt_debug_wide_walk(wide)
            if instanced:
                t1, p1, i1, b1 = scene.intersect_instanced(o, d, tmax)
                assert np.array_equal(i0, i1), (wide, int((i0 != i1).sum()))
            else:
                t1, p1, b1 = scene.intersect(o, d, tmax)
            occ1 = scene.occluded(o, d, tmax)
            assert np.array_equal(p0, p1), (wide, int((p0 != p1).sum()))
            assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(b0.view(np.uint32), b1.view(np.uint32)), wide
            assert np.array_equal(occ0, occ1), (wide, int((occ0 != occ1).sum()))
    finally:
        hprt.lib.hprt_debug_wide_walk(-1)
    assert 0.02 < (p0 >= 0).mean() < 0.98 and 0.02 < occ0.mean() < 0.98


def test_walks_agree_when_leaf_boxes_do_not_contain_their_triangles(hprt):
    """What decides whether a primitive is tested is its LEAF'S box as the tree holds it, not the primitive's extent.  A caller-filled
    HprtSceneDesc whose BVH was built over bounds shrunk to 40 % of every triangle's extent (legal input: hprt_scene_create takes the node
    array as given) makes that visible: most hits of the plain geometry are now outside their leaf's box, and both walks must drop exactly
    those — the binary walk at the parent's child test, the wide walk at the leaf's own exact test (closest hit: before the triangles count;
    any hit: when a triangle reports a hit, the path a consistent tree almost never takes).  No oracle here: the two product walks against
    each other, and against the unshrunk tree as evidence that the boxes did bite."""
    import ctypes as C
    rng = np.random.default_rng(5)
    n_tri = 3000
    c = rng.uniform(-1, 1, (n_tri, 1, 3)); e = rng.normal(size=(n_tri, 3, 3)) * rng.uniform(0.02, 0.25, (n_tri, 1, 1))
    P = (c + e).astype(np.float32).reshape(-1, 3); idx = np.arange(3 * n_tri, dtype=np.int32).reshape(-1, 3)
    tri = P[idx]
    lo, hi = tri.min(axis=1), tri.max(axis=1)
    mid = (lo + hi) * np.float32(0.5)
    shrink = lambda f: ((mid + (lo - mid) * np.float32(f)).astype(np.float32), (mid + (hi - mid) * np.float32(f)).astype(np.float32))
    sh = hprt.ShapeDesc(); sh.kind = 0; sh.material = 0; sh.area_light = -1
    sh.n_tris = n_tri; sh.n_verts = 3 * n_tri; sh.indices = idx.ctypes.data; sh.P = P.ctypes.data
    mat = hprt.MaterialDesc(); mat.type = 0; mat.Kd[:] = [.5, .5, .5]; mat.kd_texture = mat.ks_texture = -1
    n = 300000
    o = rng.uniform(-1.6, 1.6, (n, 3)).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 10, 0] = 0.0; d[n // 10: n // 5, 2] = -0.0
    tmax = np.full(n, np.inf, np.float32); tmax[n // 2:] = rng.uniform(0.1, 3.0, n - n // 2).astype(np.float32)
    hprt.lib.hprt_debug_wide_walk.argtypes = [C.c_int]
    res = {}
    try:
        for f in (1.0, 0.4):
            bmin, bmax = shrink(f)
            bvh = hprt.Bvh.from_bounds(bmin, bmax)
            nodes, order = bvh.arrays()
            desc = hprt.SceneDesc()
            desc.nodes = nodes.ctypes.data; desc.n_nodes = nodes.shape[0]; desc.prim_order = order.ctypes.data; desc.n_prims = order.shape[0]
            shapes = (hprt.ShapeDesc * 1)(sh); mats = (hprt.MaterialDesc * 1)(mat)
            desc.shapes = shapes; desc.n_shapes = 1; desc.materials = mats; desc.n_materials = 1
            scene = hprt.Scene.from_desc(desc)
            for wide in (1, 0):
                hprt.lib.hprt_debug_wide_walk(wide)
                t, p, b = scene.intersect(o, d, tmax)
                creation = np.where(p >= 0, order.astype(np.int64)[np.maximum(p, 0)], -1)      # (creation-order triangle numbers: the two trees order differently)
                res[(f, wide)] = (t, creation, b, scene.occluded(o, d, tmax))
            del scene
    finally:
        hprt.lib.hprt_debug_wide_walk(-1)
    for f in (1.0, 0.4):
        for a, b_ in zip(res[(f, 1)], res[(f, 0)]):
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b_).view(np.uint8)), f
    full, cut = res[(1.0, 1)], res[(0.4, 1)]
    assert (full[1] >= 0).mean() > 0.3
    assert (cut[1] >= 0).sum() < 0.8 * (full[1] >= 0).sum() and cut[3].sum() < 0.8 * full[3].sum()      # the shrunk boxes hide hits the geometry has
    assert not (cut[3] & ~full[3]).any()                                                                  # and never add one
