"""bench.py — Mrays/s of the wavefront path tracer on BASELINE.json's config[1]:
killeroo-simple (66,532 triangles + 1 sphere emitter), 700x700, Halton 256 spp,
PathIntegrator maxdepth 5, Accelerator "bvh", on N MI355X.

A STEP is one complete Render(): every camera sample of the frame traced through the
wavefront kernels and folded into the film (inputs — scene, BVH, sampler tables — are
resident in HBM before the timed region).  With N > 1 the image's 16x16 tiles are dealt
round-robin to the ranks and the sample count scales with N (spp = 256*N, weak scaling:
each GPU does the work of the single-GPU frame); the per-rank films are summed onto
rank 0 with one RCCL reduce over xGMI inside the timed step.

Rays = closest-hit + shadow rays, as the reference counts them (core/scene.cpp:40-55).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel = k_trace<closest>; achieved = algorithmic bytes
                (32 B per BVH node fetched + 48 B per triangle test + 28 B ray read +
                20 B hit write) / HIP-event time inside that kernel's
                launches, against the 8 TB/s HBM3E peak.
  cpu_baseline  the oracle (CPU port of the same path) on all host threads, on a
                bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FIXTURE = os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per GPU (BASELINE config[1]: 256)")
    ap.add_argument("--spp-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=128)
    ap.add_argument("--no-trace-all", action="store_true", help="skip the extra (untimed for value) frames that trace every reference ray; profiling runs use it")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug: all ranks share cuda:0 and the film is reduced with gloo (checks the N>1 control flow on a 1-GPU box)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    hprt = importlib.import_module("thesis-pbrt-v3_amd")
    tiles = importlib.import_module("thesis-pbrt-v3_amd.tiles")
    model = hprt.Model.load(FIXTURE)
    bvh = hprt.Bvh(model)
    scene = hprt.Scene(model, bvh, device=dev.index)
    opt = model.options.copy()
    opt.spp = args.spp * max(1, world)                # weak scaling: per-GPU work constant
    x0, y0, x1, y1 = opt.film_bounds()
    W, H = x1 - x0, y1 - y0
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)   # device memory via torch: plumbing only
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(count_work=False, gather=True, count_traced=True, trace_all=False):
        # count_traced: the counting pass counts the rays the timed passes trace (a plain render does not trace a BSDF-sampled
        # light ray that provably cannot reach its emitter; include/hprt.h, HPRT_RENDER_COUNT_TRACED)
        _, st = scene.render(opt, spp_chunk=args.spp_chunk, count_work=count_work, count_traced=count_work and count_traced,
                             trace_all=trace_all, film_ptr=film.data_ptr(), stream=stream, **tiles.shard(rank, max(1, world)))
        if dist is not None and gather:
            tiles.gather_film(film, dist, dst=0)             # Film tiles -> rank 0 (RCCL over xGMI); addends are disjoint
        return st

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        cdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([sum(s["rays"] + s["shadow_rays"] for s in stats), sum(s["camera_rays"] for s in stats)],
                           dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_rays, total_samples = float(tot[0].item()), float(tot[1].item())
    else:
        total_rays = float(sum(s["rays"] + s["shadow_rays"] for s in stats))
        total_samples = float(sum(s["camera_rays"] for s in stats))

    # ---- roofline of the dominant kernel (k_trace<closest hit>), rank 0's launches ----
    roofline = None
    cpu_baseline = None
    if rank == 0:
        st_c = step(count_work=True, gather=False)   # untimed, rank-local counting pass: V (nodes fetched), T (primitive tests)
        st_ref = step(count_work=True, gather=False, count_traced=False)   # untimed: the reference's full ray set (what the CPU baseline traces)
        torch.cuda.synchronize(dev)
        ref_over_traced = (st_ref["rays"] + st_ref["shadow_rays"]) / max(1, st_c["rays"] + st_c["shadow_rays"])
        # for comparison (untimed for `value`): the same frame with every ray of the reference traced (HPRT_RENDER_TRACE_ALL)
        trace_all_info = None
        if world == 1 and not args.no_trace_all:
            torch.cuda.synchronize(dev); ta0 = time.perf_counter()
            ta = [step(gather=False, trace_all=True) for _ in range(args.steps)]
            torch.cuda.synchronize(dev); ta_sec = time.perf_counter() - ta0
            trace_all_info = {"ms_per_step": round(1e3 * ta_sec / max(1, args.steps), 2),
                              "mrays_per_s": round(sum(s["rays"] + s["shadow_rays"] for s in ta) / ta_sec / 1e6, 2)}
        ext_rays = sum(s["extend_rays"] for s in stats)
        ext_sec = sum(s["extend_seconds"] for s in stats)
        ext_launches = sum(s["extend_launches"] for s in stats)
        v_per_ray = st_c["nodes_fetched"] / max(1, st_c["rays"])
        t_per_ray = (st_c["tri_tests"] + st_c["sphere_tests"]) / max(1, st_c["rays"])
        bytes_per_ray = 32.0 * v_per_ray + 48.0 * t_per_ray + 28.0 + 20.0
        achieved = ext_rays * bytes_per_ray / max(ext_sec, 1e-12) / 1e9
        # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 cannot run inside this
        # process): tools/traffic_passes.sh on this same command, summary committed under profiles/.
        traffic, traffic_from = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if world == 1 and args.spp == 256 and os.path.exists(tpath):
            try:
                kernels = json.load(open(tpath))["kernels"]
                tk = next((v for k, v in kernels.items() if k.startswith("k_trace<false, 0")), None)   # closest hit, plain (not the counting pass)
                if tk:
                    traffic, traffic_from = round(tk["hbm_bytes_per_launch"]), "profiles/r01_traffic.json"
            except Exception:
                pass
        roofline = {
            "bound": "hbm", "kernel": "k_trace<closest>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_from": traffic_from,
            "algorithmic_bytes_per_launch": round(ext_rays * bytes_per_ray / max(1, ext_launches)),
            "bytes_per_ray": round(bytes_per_ray, 1), "nodes_fetched_per_ray": round(v_per_ray, 3),
            "prim_tests_per_ray": round(t_per_ray, 3), "launches": int(ext_launches),
            "avg_launch_ms": round(1e3 * ext_sec / max(1, ext_launches), 4),
            "kernel_mrays_per_s": round(ext_rays / max(ext_sec, 1e-12) / 1e6, 1),
            "occluded_kernel_mrays_per_s": round(sum(s["occluded_rays"] for s in stats) / max(sum(s["occluded_seconds"] for s in stats), 1e-12) / 1e6, 1),
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc   # the oracle: CPU port of the same path (checker / baseline only)
            oracle = orc.OracleScene(FIXTURE)
            threads = os.cpu_count() or 1
            _, _, c, sec, nt = oracle.render(spp=args.cpu_spp, threads=threads)
            cpu_baseline = {
                "value": round((c["rays"] + c["shadow_rays"]) / sec / 1e6, 3), "unit": "Mrays/s", "cores": int(nt), "kind": "port",
                "sample": "killeroo-simple 700x700 at %d spp (of the %d spp workload), tile loop only" % (args.cpu_spp, args.spp),
                "msamples_per_s": round(c["camera_rays"] / sec / 1e6, 3), "seconds": round(sec, 2),
            }
    if dist is not None:
        dist.barrier()
    if rank == 0:
        value = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "killeroo-simple (66,532 tris + sphere light) 700x700, halton %d spp/GPU, path maxdepth 5, bvh" % args.spp,
                       "spp_total": int(opt.spp), "tiles": "16x16 round-robin over %d GPU(s)" % max(1, world), "parallelism": "tile-dp%d" % max(1, world)},
            "msamples_per_s": round(total_samples / elapsed / 1e6, 3),
            # rays actually traced.  The reference's PathIntegrator traces more: its BSDF-sampled light rays that provably cannot
            # reach the emitter are answered without a trace here (same film, bit for bit) and are NOT counted in `value`.
            "rays_per_step": int(total_rays / max(1, args.steps)),
            "reference_rays_per_step": int(total_rays * ref_over_traced / max(1, args.steps)),
            "with_every_reference_ray_traced": trace_all_info,
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if cpu_baseline:
            # same frame on both sides: ratio of frame rates (the CPU port traces the reference's full ray set)
            out["gpu_over_cpu"] = round(out["msamples_per_s"] / cpu_baseline["msamples_per_s"], 1)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
