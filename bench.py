"""bench.py — Mrays/s + Msamples/s of the wavefront path tracer on N MI355X (BASELINE.json's metric).

HEADLINE workload (config.workload) = BASELINE.json configs[2], the largest single-GPU configuration:
"Sponza (~260k tris) 1024 spp".  The reference repository ships only the scene's header (scenes/sponza:1-17; its
geometry and textures are git-ignored), so the geometry is a STAND-IN, labelled as such everywhere: the procedural
312 k-triangle atrium of tools/scene_gen.py (two storeys of arcades, tessellated columns and arches, curtains; matte and
plastic; one point light as scenes/sponza has), 700x700, Halton 1024 spp, PathIntegrator maxdepth 5, Accelerator "bvh".
Secondary workloads ride in the same JSON line (`secondary`): configs[1] killeroo-simple 256 spp (the reference's own asset,
baked), configs[3]'s class — the reference's living-room meshes at 1280x720 with the image textures it ships, at that
configuration's 2,048 spp — and configs[4] as SURVEY.md §8(d)-5 defines it (the killeroo mesh x 301 ObjectInstances, 10.01 M
triangles) at its 4,096 spp; each with its own `roofline` and `cpu_baseline`.

A STEP is one complete Render() of the headline frame: every camera sample traced through the wavefront kernels and
folded into the film; scene, BVH and sampler tables are resident in HBM before the timed region.  With N > 1 the
image's 16x16 tiles are dealt round-robin to the ranks at FIXED total spp (strong scaling, the metric's "at fixed spp";
--weak multiplies spp by N instead) and the per-rank films are merged onto rank 0 inside the timed step by
hprt_film_gather: one RCCL reduce over xGMI plus the ordered merge of the cross-tile records (csrc/capi_gather.hip).

`python bench.py --gpus N` without WORLD_SIZE in the environment starts N fresh rank processes itself (before this
process touches the GPU), watches ALL of them (the first non-zero exit ends the run at once) and relays rank 0's line; under
torch.distributed.run it reads RANK/LOCAL_RANK/WORLD_SIZE.  Every rank checks that its GPU exists before it enters a collective.
`n_gpus` is the size the RCCL communicator reports, never the flag (a --rehearse-on-one-gpu run says 1, with `world_size` ranks;
if the library's communicator cannot be created the films are merged through torch.distributed's RCCL group instead, `n_gpus` is
the number of distinct devices the ranks sit on, and `config.film_merge` says what happened).

Rays = closest-hit + shadow rays, as the reference counts them (core/scene.cpp:40-55), TRACED rays only.

Extra objects in the line:
  roofline      dominant kernel = the closest-hit walk that ran (`kernel`): k_walk4<closest>, the leaf-exact four-wide walk of plain renders
                (DESIGN.md section 4), or k_trace<closest>, the binary walk (HPRT_WIDE_WALK=0).  `walk` = what k_walk4 fetched per ray, counted by ONE
                untimed render of its own phase-profile variant: 64-byte wide records stepped, primitive records tested, 16-byte requests.
                `frac` = MEASURED HBM traffic of that kernel (FETCH_SIZE / WRITE_SIZE PMC passes of
                the committed summary profiles/rNN_counters.json, per launch) / this run's HIP-event launch time / the 8 TB/s HBM3E
                peak: at most 1 by construction; `peak_measured` = a float4 stream copy run inside this process (what a stream kernel
                reaches on THIS box) with `frac_of_measured_peak`.  `bound` names the largest of three measured fractions
                (`bound_candidates`), each of a peak and so at most 1: "hbm" (`frac`); "valu_issue" = SQ_INSTS_VALU * 2 / (1,024 SIMDs *
                kernel cycles), cycles = GRBM_GUI_ACTIVE / 8 — CDNA4's SIMD-32 issues one wave64 instruction per two clocks — with
                `lane_utilisation` and `useful_lane_frac` = issue x lanes; "l1_gather" (`gather`) = the 64-byte records the kernel
                fetches per second (k_walk4: its 16-byte requests / 4, from `walk`; k_trace: V / 2 + 0.75 T per ray from the counting pass; x its ray rate) over the
                rate at which THIS GPU serves dependent per-lane 64-byte gathers that all hit L1 (k_gather_probe, run inside this process
                in k_trace's launch shape with no arithmetic at all: ~225 G records/s); the same probe over a table of the BVH's size with
                a BVH-like pick rides along (`bvh_like`, ~186 G on the atrium).  The binary walk ran at 0.82 of that ceiling; the wide walk asks for half the records
                and sits at 0.58, with VALU issue at 0.50 and HBM at 0.36: `bound` is the largest of the three, none of them a wall by itself.
                The summary is stamped with the sha256 of the code objects it was measured on; `counters_stale` says when the loaded
                library's differ (then `frac` falls back to the summary's own launch time and the flag tells).  The contract's model
                figure — algorithmic bytes (32 B per BVH node fetched + 48 B per primitive test + 28 B ray + 20 B hit; V
                and T counted by the kernel) / kernel time — rides in `roofline.algorithmic` labelled as what it is: the BVH is served by
                L1 / L2 / MALL, so that rate exceeds the HBM peak and is NOT a roofline fraction.
  cpu_baseline  the oracle (CPU port of the same path, kind "port") on the CPUs this process can really use — min(affinity, cgroup
                quota): the box shows 256 logical CPUs and grants 16 — on a bounded sample of the same frame, with the host's
                description and a thread sweep (`scaling`).  BASELINE.md §3 relates the port's speed to the reference binary's.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
N_SIMDS = 1024          # 256 CUs x 4 SIMDs


def counters_file():
    """The newest committed counter summary, profiles/rNN_counters.json (tools/counters_passes.sh + collect_profiles.py)."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counters.json")))
    return fs[-1] if fs else None


def code_object_hash(lib_path):
    """sha256 of the gfx950 code objects inside a built libhprt.so (its .hip_fatbin section): what ties a counter summary to
    the kernels that were measured.  Host-only changes of the library do not move it."""
    import hashlib, struct
    try:
        with open(lib_path, "rb") as f:
            data = f.read()
        if data[:4] != b"\x7fELF" or data[4] != 2:
            return None
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
        def sh(i):
            name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
            return name, off, size
        _, stroff, strsize = sh(shstrndx)
        for i in range(shnum):
            name, off, size = sh(i)
            end = data.index(b"\0", stroff + name)
            if data[stroff + name:end] == b".hip_fatbin":
                return hashlib.sha256(data[off:off + size]).hexdigest()
    except Exception:
        return None
    return None

WORKLOADS = {
    # name: (description, how to build the model, spp, cpu sample spp, data label)
    "atrium": ("Sponza-class atrium STAND-IN (311,728 tris, matte+plastic, point light; the reference ships no Sponza geometry) 700x700, "
               "halton %d spp, path maxdepth 5, bvh", 1024, 64, "synthetic (procedural stand-in geometry, tools/scene_gen.py)"),
    "killeroo-simple": ("killeroo-simple (66,532 tris + sphere area light) 700x700, halton %d spp, path maxdepth 5, bvh", 256, 128,
                        "reference asset scenes/killeroo-simple, baked (tests/golden/killeroo_simple.hprt)"),
    "living-room": ("living room (143,163 tris of the reference's scenes/living-room meshes with its matte / OrenNayar / substrate / metal / mirror / glass / uber materials and the two image textures it ships — picture8.tga on the painting, leaf.tga on the leaves' Kd and opacity; the two stripped wood maps as constants, point light for its missing sky map) "
                    "1280x720, halton %d spp, path maxdepth 5, bvh", 2048, 32, "reference meshes, baked (tests/golden/living_room.hprt); BASELINE configs[3]'s class (the conference-room blob is stripped from the reference) at its 2,048 spp; image textures picture8.tga + leaf.tga (Kd and uber opacity) as the scene binds them"),
    # BASELINE configs[4] as SURVEY.md §8(d)-5 defines it: TransformedPrimitive instancing (core/primitive.cpp:77-102, core/api.cpp:1778-1820)
    "instanced-10m": ("10.01 M-triangle instanced scene: the reference's killeroo mesh (33,264 tris, one object definition) x 301 ObjectInstances on a jittered 7x7x7 "
                      "lattice (PCG32 sequence 5), ground quad, distant light; two-level BVH; 700x700, halton %d spp, path maxdepth 5, bvh", 4096, 8,
                      "reference mesh (baked, tests/golden/killeroo.hprt) instanced by tools/scene_gen.py:instanced_killeroo"),
}
SECONDARY = ("killeroo-simple", "living-room", "instanced-10m")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="atrium", choices=sorted(WORKLOADS), help="headline workload (default: BASELINE configs[2])")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of the headline workload (0: the configuration's own)")
    ap.add_argument("--spp-chunk", type=int, default=0)
    ap.add_argument("--weak", action="store_true", help="weak scaling: spp x N (default: fixed total spp, strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-trace-all", action="store_true", help="skip the comparison frames that trace every reference ray")
    ap.add_argument("--profile-step", action="store_true",
                    help="profiling runs: ONE plain step of the headline workload and nothing else (no warm-up, counting passes, "
                         "secondary workloads or CPU baseline), so that a rocprofv3 summary of the process is the summary of a step")
    ap.add_argument("--dump-film", default="", help="rank 0 saves the merged film of the last headline step (numpy [H,W,4]); tests use it")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0 and the films are merged through gloo (RCCL refuses two ranks on one device): "
                         "the N>1 control flow on a 1-GPU box")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# host CPUs: what the CPU baseline can really use
# ---------------------------------------------------------------------------------------------
def _cgroup_cpu_quota():
    """CPUs' worth of quota the cgroup hierarchy of this process allows (None: unlimited), and where it was read."""
    best, where = None, None
    def consider(q, w):
        nonlocal best, where
        if q is not None and (best is None or q < best):
            best, where = q, w
    try:      # cgroup v2: cpu.max = "<quota|max> <period>" at every level from this process's group up to the root
        rel = ""
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
        d = os.path.normpath("/sys/fs/cgroup/" + rel.lstrip("/"))
        while d.startswith("/sys/fs/cgroup"):
            f = os.path.join(d, "cpu.max")
            if os.path.exists(f):
                q, per = open(f).read().split()[:2]
                if q != "max":
                    consider(float(q) / float(per), f)
            if d == "/sys/fs/cgroup":
                break
            d = os.path.dirname(d)
    except Exception:
        pass
    try:      # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            consider(q / per, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us")
    except Exception:
        pass
    return best, where


def host_cpu_info():
    """logical CPUs the OS shows, CPUs this process may run on (affinity), the cgroup's CPU quota, and the CPU model.
    `usable` = what a thread pool can actually get: min(affinity, quota rounded up)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = logical
    quota, where = _cgroup_cpu_quota()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except Exception:
        pass
    usable = affinity if quota is None else max(1, min(affinity, int(-(-quota // 1))))
    return {"logical_cpus": logical, "affinity_cpus": affinity, "cgroup_cpu_quota": None if quota is None else round(quota, 2),
            "cgroup_quota_from": where, "usable_cpus": usable, "cpu_model": model}


def sweep_threads(info):
    return sorted({t for t in (1, 8, 32, info["usable_cpus"], info["affinity_cpus"]) if 1 <= t <= info["logical_cpus"]})


# ---------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started before this process makes any GPU call
# ---------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Starts args.gpus fresh rank processes and relays rank 0's stdout.  The parent watches ALL of them: the first child that
    exits non-zero ends the run — the others are terminated (they would otherwise sit in init_process_group or in the first
    collective until somebody's time limit) and the parent exits non-zero at once.  Rank r's stderr is relayed line by line,
    tagged "[rank r]".  Children are only ever started fresh; a process that touched the GPU is never re-executed."""
    import threading
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, out_chunks, threads = [], [], []
    def relay_err(r, pipe):
        for line in iter(pipe.readline, b""):
            sys.stderr.write("[rank %d] %s" % (r, line.decode("utf-8", "replace"))); sys.stderr.flush()
    def collect_out(pipe):
        for chunk in iter(lambda: pipe.read(65536), b""):
            out_chunks.append(chunk)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(p)
        t = threading.Thread(target=relay_err, args=(r, p.stderr), daemon=True); t.start(); threads.append(t)
        if r == 0:
            t = threading.Thread(target=collect_out, args=(p.stdout,), daemon=True); t.start(); threads.append(t)
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.1)
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; terminating the other ranks\n" % failed)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
    for t in threads:
        t.join(timeout=5.0)
    sys.stdout.write(b"".join(out_chunks).decode())
    sys.stdout.flush()
    if failed is not None:
        return abs(failed[1]) or 1
    return 0


# ---------------------------------------------------------------------------------------------
def build_model(hprt, name):
    if name in ("atrium", "instanced-10m"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scene_gen
        text, _ = scene_gen.atrium(1.0) if name == "atrium" else scene_gen.instanced_killeroo(os.path.join(GOLDEN, "killeroo.hprt"))
        d = tempfile.mkdtemp(prefix="hprt_bench_")
        path = os.path.join(d, name + ".pbrt")
        with open(path, "w") as f:
            f.write(text)
        return hprt.Model.parse(path)
    return hprt.Model.load(os.path.join(GOLDEN, {"killeroo-simple": "killeroo_simple.hprt", "living-room": "living_room.hprt"}[name]))


class Workload:
    """One scene resident on this rank's GPU + its film buffer."""

    def __init__(self, hprt, tiles, torch, name, spp, dev, rank, world):
        self.name = name
        self.model = build_model(hprt, name)
        self.bvh = hprt.Bvh(self.model)
        self.scene = hprt.Scene(self.model, self.bvh, device=dev.index)
        self.opt = self.model.options.copy()
        self.opt.spp = spp
        x0, y0, x1, y1 = self.opt.film_bounds()
        self.W, self.H = x1 - x0, y1 - y0
        self.film = torch.zeros((self.H, self.W, 4), dtype=torch.float32, device=dev)   # device memory via torch: plumbing only
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.shard = tiles.shard(rank, world)
        self.sharded = world > 1

    def render(self, spp_chunk=0, **kw):
        _, st = self.scene.render(self.opt, spp_chunk=spp_chunk, film_ptr=self.film.data_ptr(), stream=self.stream,
                                  export_foreign=self.sharded, **self.shard, **kw)
        return st


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))          # nothing above touched the GPU
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to label one as the other\n" % (args.gpus, world))
        sys.exit(2)

    import datetime
    import numpy as np
    import torch

    # ---- pre-flight, before any collective can be entered: a rank that cannot run must leave with a non-zero code NOW, so that
    # the launcher (launch_ranks, or torch.distributed.run) ends the job instead of letting its peers wait for it ----
    if os.environ.get("HPRT_TEST_FAIL_RANK") == str(rank):      # test hook: this rank dies during start-up
        sys.stderr.write("bench.py: HPRT_TEST_FAIL_RANK=%d: failing on purpose before the process group\n" % rank)
        sys.exit(7)
    n_dev = torch.cuda.device_count()                           # (does not initialise the GPU)
    need = 1 if (world == 1 or args.rehearse_on_one_gpu) else local_rank + 1
    if n_dev < need:
        sys.stderr.write("bench.py: rank %d needs cuda:%d but this node shows %d GPU(s)\n" % (rank, need - 1, n_dev))
        sys.exit(3)
    pg_timeout = datetime.timedelta(seconds=int(os.environ.get("HPRT_BENCH_PG_TIMEOUT", "180")))

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo", timeout=pg_timeout)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    hprt = importlib.import_module("thesis-pbrt-v3_amd")
    tiles = importlib.import_module("thesis-pbrt-v3_amd.tiles")

    # ---- the film gather's communicator: RCCL through the C ABI (hprt_comm_*); the id travels over torch.distributed ----
    comm, transport, rccl_ranks = None, "none (single GPU)", None
    n_gpus = 1
    if world > 1 and not args.rehearse_on_one_gpu:
        err = ""
        try:
            idt = torch.zeros(hprt.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(hprt.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            # ncclCommInitRank blocks until every rank has joined: bound it, a peer may have died since the broadcast above
            import threading
            box = {}
            def make():
                try:
                    box["comm"] = hprt.Comm(bytes(idt.cpu().numpy().tobytes()), rank, world, device=dev.index)
                except Exception as e:
                    box["err"] = e
            th = threading.Thread(target=make, daemon=True); th.start(); th.join(timeout=pg_timeout.total_seconds())
            if th.is_alive():
                sys.stderr.write("bench.py: rank %d: hprt_comm_create did not return within %d s; giving up\n" % (rank, pg_timeout.total_seconds()))
                sys.stderr.flush()
                os._exit(4)      # (the thread sits inside RCCL: nothing to clean up from here)
            if "err" in box:
                raise box["err"]
            comm = box["comm"]
            rccl_ranks = comm.info()["n_ranks"]
        except Exception as e:      # e.g. the library's own communicator cannot be set up in this environment
            comm, rccl_ranks, err = None, None, "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            n_gpus, transport = rccl_ranks, "RCCL (hprt_film_gather: ncclReduce + grouped send/recv)"
        else:
            # Every rank still owns a GPU and torch.distributed's own RCCL group works (the all_reduce above ran on it): merge
            # the films through it (tiles.gather_film: reduce + ordered record merge on the root) and say so in the line.
            comm, rccl_ranks = None, None
            devs = [None] * world
            dist.all_gather_object(devs, int(dev.index))
            n_gpus = len(set(devs))
            transport = "torch.distributed RCCL group (tiles.gather_film) — hprt_comm_create failed: " + (err or "on another rank")
            if rank == 0:
                sys.stderr.write("bench.py: " + transport + "\n")
    elif world > 1:
        transport = "gloo rehearsal on one GPU (tiles.gather_film)"      # world ranks on ONE card: n_gpus stays 1

    def merge(w):
        """Film::MergeFilmTile across ranks; part of the timed step."""
        if comm is not None:
            comm.film_gather(w.scene, w.film.data_ptr(), w.W * w.H, root=0, stream=w.stream)
        elif dist is not None:
            tiles.gather_film(w.film, dist, dst=0, records=w.scene.film_records())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(xs):
        if dist is None:
            return [float(x) for x in xs]
        t = torch.tensor(xs, dtype=torch.float64, device=torch.device("cpu") if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def timed(w, steps, warmup):
        for _ in range(warmup):
            w.render(args.spp_chunk); merge(w)
        barrier()
        t0 = time.perf_counter()
        stats = []
        for _ in range(steps):
            stats.append(w.render(args.spp_chunk)); merge(w)
        barrier()
        elapsed = reduce_max(time.perf_counter() - t0)
        rays, samples = reduce_sum([sum(s["rays"] + s["shadow_rays"] for s in stats), sum(s["camera_rays"] for s in stats)])
        return elapsed, rays, samples, stats

    def cpu_port(name, spp_sample, spp_full, sweep=False):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc   # the oracle: CPU port of the same path (checker / baseline only)
        d = tempfile.mkdtemp(prefix="hprt_bench_")
        baked = os.path.join(d, name + ".hprt")
        build_model(hprt, name).save(baked)
        oracle = orc.OracleScene(baked)
        # threads = the CPUs this process can really get: os.cpu_count() shows the whole machine, the box's lease (CPU affinity
        # and / or a cgroup quota) is usually a slice of it, and oversubscribing a quota only buys throttling
        cpu = host_cpu_info()
        threads = cpu["usable_cpus"]
        _, _, c, sec, nt = oracle.render(spp=spp_sample, threads=threads)
        out = {"value": round((c["rays"] + c["shadow_rays"]) / sec / 1e6, 3), "unit": "Mrays/s", "cores": int(nt), "kind": "port",
               "sample": "%s at %d spp (of the %d spp workload), whole frame, tile loop only" % (name, spp_sample, spp_full),
               "msamples_per_s": round(c["camera_rays"] / sec / 1e6, 3), "seconds": round(sec, 2),
               "mrays_per_s_per_thread": round((c["rays"] + c["shadow_rays"]) / sec / 1e6 / max(1, nt), 4), "host": cpu}
        if sweep:      # how the port scales on this host: the same frame per thread count, about a second each (long enough for a cgroup quota to bite)
            out["scaling"] = []
            for t in sweep_threads(cpu):
                _, _, c1, s1, n1 = oracle.render(spp=max(1, min(16, t // 4)), threads=t)
                r1 = (c1["rays"] + c1["shadow_rays"]) / s1 / 1e6
                out["scaling"].append({"threads": int(n1), "mrays_per_s": round(r1, 3), "per_thread": round(r1 / max(1, n1), 4), "seconds": round(s1, 2)})
        return out

    # =========================================================================================
    # headline workload
    # =========================================================================================
    desc, spp_cfg, cpu_spp, data_label = WORKLOADS[args.workload]
    spp = (args.spp or spp_cfg) * (world if args.weak else 1)
    head = Workload(hprt, tiles, torch, args.workload, spp, dev, rank, world)

    if args.profile_step:
        barrier()
        t0 = time.perf_counter(); st = head.render(args.spp_chunk); merge(head); barrier()
        if rank == 0:
            print(json.dumps({"profile_step": args.workload, "spp": spp, "ms": round(1e3 * (time.perf_counter() - t0), 2), "closest_mrays_s": round(st["extend_rays"] / max(st["extend_seconds"], 1e-9) / 1e6), "any_mrays_s": round(st["occluded_rays"] / max(st["occluded_seconds"], 1e-9) / 1e6),
                              "rays": st["rays"] + st["shadow_rays"]}))
        if dist is not None:
            dist.destroy_process_group()
        return

    elapsed, total_rays, total_samples, stats = timed(head, args.steps, args.warmup)
    if rank == 0 and args.dump_film:
        np.save(args.dump_film, head.film.cpu().numpy())

    def kernel_figures(w, stats):
        """rank 0: counting passes (untimed) + the roofline object of k_trace<closest> for workload w"""
        st_c = w.render(count_work=True, count_traced=True)     # V (nodes fetched), T (primitive tests) of the rays the timed steps trace
        st_ref = w.render(count_work=True)                      # the reference's full ray set (what the CPU baseline traces)
        torch.cuda.synchronize(dev)
        ref_over_traced = (st_ref["rays"] + st_ref["shadow_rays"]) / max(1, st_c["rays"] + st_c["shadow_rays"])
        ext_rays = sum(s["extend_rays"] for s in stats); ext_sec = sum(s["extend_seconds"] for s in stats)
        ext_launches = sum(s["extend_launches"] for s in stats)
        v_per_ray = st_c["nodes_fetched"] / max(1, st_c["rays"])
        t_per_ray = (st_c["tri_tests"] + st_c["sphere_tests"]) / max(1, st_c["rays"])
        bytes_per_ray = 32.0 * v_per_ray + 48.0 * t_per_ray + 28.0 + 20.0
        achieved = ext_rays * bytes_per_ray / max(ext_sec, 1e-12) / 1e9
        avg_ms = 1e3 * ext_sec / max(1, ext_launches)
        # The contract's model figure: algorithmic bytes / kernel time.  The BVH is served by L2 / MALL, so it is a
        # request rate, not an HBM utilisation, and it is NOT what `frac` reports: `frac` is filled from the measured HBM traffic
        # of the counter summary below (and stays null when there is none for this build).
        # which walk the timed steps took, and — for the leaf-exact wide walk (k_walk4) — what it fetched: one untimed render of the
        # phase-profile variant counts the lane-steps (one 64-byte wide record each: four 16-byte requests) and the leaf iterations
        # (one 48-byte primitive record each: three requests; leaves that are not one triangle add their 32-byte box)
        import ctypes as C
        wide = False
        try:
            fn = hprt.lib.hprt_debug_scene_walk; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
            wide = bool(fn(w.scene._h))
        except Exception:
            wide = False
        walk = None
        if wide:
            try:
                mode = hprt.lib.hprt_debug_trace_profile_mode; mode.argtypes = [C.c_int]; mode.restype = C.c_int
                prof = hprt.lib.hprt_debug_trace_profile; prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; prof.restype = C.c_int
                buf = (C.c_ulonglong * 32)()
                mode(1); prof(buf, 1)
                opt_save = w.opt.spp
                w.opt.spp = max(1, min(opt_save, 16))      # (the per-ray averages do not depend on the sample count; the profile variants are slow)
                st_p = w.render()
                torch.cuda.synchronize(dev)
                w.opt.spp = opt_save
                prof(buf, 1); mode(-1)
                v = [int(x) for x in buf]
                nr = max(1, st_p["rays"])
                walk = {"kernel": "k_walk4<closest>", "records_per_ray": round(v[6] / nr, 3), "leaf_iterations_per_ray": round(v[8] / nr, 3),
                        "leaf_boxes_tested_per_ray": round(v[11] / nr, 3), "leaf_boxes_passed_per_ray": round(v[12] / nr, 3),
                        "lanes_per_record_step": round(v[6] / max(1, v[5]), 1), "lanes_per_leaf_iteration": round(v[8] / max(1, v[7]), 1),
                        "requests_per_ray": round((4.0 * v[6] + 3.0 * v[8]) / nr, 2),
                        "any_hit": {"records_per_ray": round(v[16 + 6] / max(1, st_p["shadow_rays"]), 3), "leaf_iterations_per_ray": round(v[16 + 8] / max(1, st_p["shadow_rays"]), 3)},
                        "what": "one untimed render of the phase-profile variant: 64-byte wide records stepped and primitive records tested per closest-hit ray; "
                                "requests = 4 x records + 3 x leaf iterations (16 bytes each)"}
            except Exception as e:
                walk = {"error": repr(e)}
        roof = {
            "bound": None, "kernel": "k_walk4<closest>" if wide else "k_trace<closest>", "walk": walk, "achieved": None, "peak": HBM_PEAK_GBS, "peak_measured": None, "unit": "GB/s",
            "frac": None, "traffic": None,
            "algorithmic": {"bytes_per_ray": round(bytes_per_ray, 1), "nodes_fetched_per_ray": round(v_per_ray, 3), "prim_tests_per_ray": round(t_per_ray, 3),
                            "bytes_per_launch": round(ext_rays * bytes_per_ray / max(1, ext_launches)), "rate_gbs": round(achieved, 2),
                            "note": "32 B per node fetched + 48 B per primitive test + 28 B ray + 20 B hit (SURVEY §8(d)), V and T counted by the kernel; "
                                    "served mostly by L1 / L2 / MALL: a request rate, not an HBM utilisation, and not a roofline fraction"},
            "launches": int(ext_launches), "avg_launch_ms": round(avg_ms, 4),
            "kernel_mrays_per_s": round(ext_rays / max(ext_sec, 1e-12) / 1e6, 1),
            "occluded_kernel_mrays_per_s": round(sum(s["occluded_rays"] for s in stats) / max(sum(s["occluded_seconds"] for s in stats), 1e-12) / 1e6, 1),
        }
        return roof, ref_over_traced

    # ---- on-box HBM stream bandwidth (a float4 copy of 4 GiB, far beyond the 256 MB MALL): the peak a stream kernel reaches HERE ----
    stream_gbs = None
    if rank == 0 and world == 1:
        import ctypes as C
        try:
            fn = hprt.lib.hprt_debug_stream_copy
            fn.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            best, mean = C.c_double(), C.c_double()
            if fn(dev.index, 4 << 30, 10, C.byref(best), C.byref(mean)) == 0:
                stream_gbs = {"best": round(best.value, 1), "mean": round(mean.value, 1), "what": "float4 copy, 4 GiB -> 4 GiB, read + write bytes / HIP-event time, 10 launches"}
        except Exception as e:
            stream_gbs = {"error": repr(e)}
        torch.cuda.empty_cache()
    # ---- on-box gather ceiling: dependent per-lane fetches of 64-byte records over a BVH-like pick, in k_trace<closest>'s launch
    # shape and with no arithmetic (kernels.hip, k_gather_probe); the table has as many records as the scene's BVH has child pairs ----
    gather_peak = None
    if rank == 0 and world == 1:
        import ctypes as C
        try:
            bi = head.bvh.info()
            pairs = max(256, bi["nodes"] - bi["leaves"])
            log2_records = min(26, max(8, int(np.ceil(np.log2(pairs)))))
            fn = hprt.lib.hprt_debug_gather_probe
            fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            best, mean, best2, mean2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
            # (a) 256 records = 16 KB: every fetch hits L1 — the rate no per-lane gather of 64-byte records can exceed on this GPU;
            # (b) as many records as the BVH has pairs, picked like BVH nodes: what a walk with no arithmetic and no divergence gets
            if fn(dev.index, 8, 2000, 5, C.byref(best), C.byref(mean)) == 0 and fn(dev.index, log2_records, 2000, 5, C.byref(best2), C.byref(mean2)) == 0:
                gather_peak = {"best": round(best.value, 1), "mean": round(mean.value, 1), "records": 256,
                               "bvh_like": {"best": round(best2.value, 1), "mean": round(mean2.value, 1), "records": 1 << log2_records},
                               "what": "k_gather_probe: every lane fetches its own 64-byte record (4 x b128), the next pick depends on the record read; "
                                       "256 threads, 24 KB LDS, 6 workgroups per CU (k_trace<closest>'s launch shape), no arithmetic; 5 launches.  best / mean: "
                                       "a 16 KB table (every fetch an L1 hit: the ceiling); bvh_like: a table of the BVH's size, level of a complete binary "
                                       "tree uniformly then a node of that level (a model of a walk's locality, not a ceiling)"}
        except Exception as e:
            gather_peak = {"error": repr(e)}
        torch.cuda.empty_cache()
    lib_hash = code_object_hash(hprt.LIB_PATH)

    def attach_counters(roof, workload, live):
        """Measured figures of the SAME kernels from the committed rocprofv3 summary (rocprofv3 cannot run inside this process):
        bound, HBM traffic and fraction, VALU issue fraction, lane utilisation.  The summary is stamped with the hash of the code
        objects it was taken on; when the loaded library's differs the figures are flagged stale (and `frac` is not filled)."""
        roof["peak_measured"] = stream_gbs
        if live and gather_peak and gather_peak.get("best"):
            # what the traversal kernel is actually held against: the rate at which this GPU serves its access pattern.  A child-pair
            # record is 64 bytes (two nodes of V), a triangle 48 bytes (three of the four 16-byte requests of a record)
            a = roof["algorithmic"]
            wk = roof.get("walk")
            if wk and wk.get("requests_per_ray"):
                # the wide walk: 16-byte requests counted by its own profile variant, in units of 64-byte records (four requests)
                rec_per_ray = wk["requests_per_ray"] / 4.0
                note = ("k_walk4<closest>: (4 x wide records stepped + 3 x primitive records tested) / 4 per ray x its ray rate, over the rate at which this GPU "
                        "serves per-lane 64-byte gathers (four 16-byte requests) that all hit L1; the binary walk k_trace ran at 0.82 of it, the wide walk asks for half "
                        "as many records per ray and is no longer held by this rate alone (the kernel's HBM fraction is `frac`)")
            else:
                rec_per_ray = a["nodes_fetched_per_ray"] / 2.0 + 0.75 * a["prim_tests_per_ray"]
                note = ("child-pair records (V / 2) + triangle records (0.75 T) per ray x k_trace<closest>'s ray rate, over the rate at which this "
                        "GPU serves per-lane 64-byte gathers that all hit L1 (the bound of this access pattern; the kernel's HBM fraction is `frac`)")
            ach = roof["kernel_mrays_per_s"] * 1e6 * rec_per_ray / 1e9
            roof["gather"] = {"achieved": round(ach, 1), "peak_measured": gather_peak, "unit": "Grecords/s", "frac": round(ach / gather_peak["best"], 4),
                              "records_per_ray": round(rec_per_ray, 2), "note": note}
            if workload == args.workload:      # (the BVH-sized probe was run for the headline's BVH)
                roof["gather"]["over_bvh_like_model"] = round(ach / gather_peak["bvh_like"]["best"], 4)
            else:
                roof["gather"]["peak_measured"] = {k: v for k, v in gather_peak.items() if k != "bvh_like"}
        cf = counters_file()
        if not cf:
            roof["counters_from"] = None
            return
        roof["counters_from"] = os.path.relpath(cf, ROOT)
        try:
            cj_all = json.load(open(cf))
            cj = cj_all.get(workload)
            stamp = cj_all.get("_stamp", {})
            roof["counters_commit"] = stamp.get("commit")
            roof["counters_code_object_sha256"] = stamp.get("code_object_sha256")
            roof["library_code_object_sha256"] = lib_hash
            roof["counters_stale"] = not (lib_hash and stamp.get("code_object_sha256") == lib_hash)
            if not cj:
                return
            k = cj["kernels"]
            tc = (k.get("k_walk4<closest>") or k.get("k_walk4<closest>[inst]")) if roof["kernel"].startswith("k_walk4") else (k.get("k_trace<closest>") or k.get("k_trace<closest>[inst]"))
            if tc and tc.get("hbm_bytes_per_launch") is not None:
                roof["traffic"] = round(tc["hbm_bytes_per_launch"])
                # per launch of THIS run: the summary's bytes per launch over this run's HIP-event launch time (the two runs launch
                # the same kernels on the same rays: avg_launch_ms of the summary rides along for comparison)
                ms = roof["avg_launch_ms"] if live and not roof["counters_stale"] else tc["avg_launch_ms"]
                roof["achieved"] = round(tc["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
                roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
                if stream_gbs and stream_gbs.get("best"):
                    roof["frac_of_measured_peak"] = round(roof["achieved"] / stream_gbs["best"], 4)
                roof["rocprof_avg_launch_ms"] = tc["avg_launch_ms"]
            for f in ("valu_issue_frac", "lane_utilisation", "useful_lane_frac", "wait_frac", "effective_clock_mhz"):
                if tc and f in tc:
                    roof[f] = tc[f]
            # what binds the kernel: the largest of the three measured fractions — HBM traffic over the HBM peak (counters), VALU
            # instructions over the SIMDs' issue rate (counters; one wave64 instruction per two clocks, tools/counters_summary.py), and
            # record fetches over the measured gather ceiling (live, `gather`).  Each is a fraction of a peak: none can exceed 1.
            cand = {}
            if roof.get("frac") is not None:
                cand["hbm"] = roof["frac"]
            if tc and tc.get("valu_issue_frac") is not None:
                cand["valu_issue"] = tc["valu_issue_frac"]
            if roof.get("gather"):
                cand["l1_gather"] = roof["gather"]["frac"]
            if cand:
                roof["bound"] = max(cand, key=cand.get)
                roof["bound_frac"] = cand[roof["bound"]]
                roof["bound_candidates"] = cand
            roof["per_kernel"] = {name: {f: v[f] for f in ("lane_utilisation", "valu_issue_frac", "useful_lane_frac", "wait_frac", "avg_launch_ms", "launches", "hbm_gbs", "hbm_frac", "l1_accesses_per_clk_cu", "l1_miss_rate", "l2_hit_rate") if f in v and v[f] is not None}
                                  for name, v in k.items()}
            roof["hbm_bytes_per_step"] = cj.get("hbm_bytes_per_step")
            roof["step_hbm_frac"] = cj.get("step_hbm_frac")
        except Exception as e:   # a malformed summary must not take the bench down
            roof["counters_error"] = repr(e)

    roofline = cpu_baseline = trace_all_info = None
    ref_over_traced = 1.0
    if rank == 0 and world == 1:
        roofline, ref_over_traced = kernel_figures(head, stats)
        attach_counters(roofline, args.workload, live=True)
        if not args.no_trace_all:
            torch.cuda.synchronize(dev); ta0 = time.perf_counter()
            ta = [head.render(args.spp_chunk, trace_all=True) for _ in range(args.steps)]
            torch.cuda.synchronize(dev); ta_sec = time.perf_counter() - ta0
            trace_all_info = {"ms_per_step": round(1e3 * ta_sec / max(1, args.steps), 2),
                              "mrays_per_s": round(sum(s["rays"] + s["shadow_rays"] for s in ta) / ta_sec / 1e6, 2)}

    # =========================================================================================
    # secondary workloads (same line, not the headline)
    # =========================================================================================
    secondary = []
    if not args.no_secondary:
        for name in [n for n in SECONDARY if n != args.workload] + (["atrium"] if args.workload != "atrium" else []):
            d2, spp2, cpu2, label2 = WORKLOADS[name]
            spp2 = spp2 * (world if args.weak else 1)
            del_w = Workload(hprt, tiles, torch, name, spp2, dev, rank, world)
            steps2 = max(1, min(args.steps, 3 if name == "killeroo-simple" else 2))      # (the 2,048 / 4,096 spp frames take seconds each)
            e2, r2, s2, st2 = timed(del_w, steps2, 1)
            item = None
            if rank == 0:
                item = {"workload": d2 % spp2, "data": label2, "steps": steps2, "ms_per_step": round(1e3 * e2 / steps2, 2),
                        "mrays_per_s": round(r2 / e2 / 1e6, 2), "msamples_per_s": round(s2 / e2 / 1e6, 3), "rays_per_step": int(r2 / steps2)}
                if world == 1:
                    roof2, rot2 = kernel_figures(del_w, st2)
                    attach_counters(roof2, name, live=True)
                    item["reference_rays_per_step"] = int(r2 * rot2 / steps2)
                    item["roofline"] = roof2
                    if not args.no_cpu_baseline:
                        item["cpu_baseline"] = cpu_port(name, cpu2, spp2)
                        item["gpu_over_cpu"] = round(item["msamples_per_s"] / item["cpu_baseline"]["msamples_per_s"], 1)
                secondary.append(item)
            del del_w
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_port(args.workload, cpu_spp, spp, sweep=True)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        value = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 2), "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f32", "data": data_label,
            "config": {"workload": desc % spp, "baseline_config": "configs[2]" if args.workload == "atrium" else args.workload,
                       "spp_total": int(spp), "tiles": "16x16 round-robin over %d GPU(s)" % world, "parallelism": "tile-dp%d" % world,
                       "film_merge": transport},
            "rccl_ranks": rccl_ranks, "world_size": world,
            "msamples_per_s": round(total_samples / elapsed / 1e6, 3),
            # rays actually traced.  The reference's PathIntegrator traces more: BSDF-sampled light rays that provably cannot reach
            # the emitter and the segment behind a path's last vertex are not traced here (same film, bit for bit), nor counted.
            "rays_per_step": int(total_rays / max(1, args.steps)),
            "reference_rays_per_step": int(total_rays * ref_over_traced / max(1, args.steps)) if world == 1 else None,
            "with_every_reference_ray_traced": trace_all_info,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "secondary": secondary,
        }
        if cpu_baseline:
            # same frame on both sides: ratio of frame rates (the CPU port traces the reference's full ray set)
            out["gpu_over_cpu"] = round(out["msamples_per_s"] / cpu_baseline["msamples_per_s"], 1)
        print(json.dumps(out))
    comm = None      # (ncclCommDestroy now, not at interpreter shutdown)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
