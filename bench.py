#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and s/frame of the path-traced cerealbar-synth frame (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU. Started plainly, this script spawns its N ranks itself (fresh child processes, created
before anything here touches the GPU; rank 0's JSON line is relayed, a failing rank makes the exit code non-zero).
Started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in the
environment) it is one of those ranks.

A step is one full 1920x1080 frame at 128 spp through the hot path (ray-gen, extend, shade, shadow,
film), config C3 of BASELINE.json, on the synthetic stand-in for assets/cerealBar.xml (the real geometry is
Git-LFS only; see raytracer_amd/synth.py). With N ranks the 16x16 buckets are dealt round-robin to the ranks, every
rank traces its buckets into a full-frame film in its own HBM, and the films are reduced to rank 0 over RCCL
(sum of colour/weight/count, min of z: RayCaster.cpp:1182-1212) inside the timed region. `value` counts every ray
segment traced by all ranks (primary + bounce + shadow hops) divided by the max-over-ranks wall time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tri-scale", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--max-bounces", type=int, default=64)
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline (0 = the benchmark's own --spp)")
    ap.add_argument("--group", action="store_true",
                    help="drive the N GPUs from ONE process through the C ABI's rtx_group_* (one host thread per GPU, films merged "
                         "with hipMemcpyPeer) instead of one RCCL rank per GPU: no torch, no RCCL in the measured process")
    ap.add_argument("--group-devices", default="", help="device ids of the group members, comma separated (default 0..N-1; an id may "
                                                        "repeat: rehearsal of the N-member code path on fewer GPUs)")
    ap.add_argument("--legacy", action="store_true",
                    help="legacy shading mode (SHADING_USE_LEGACY=1) instead of the headline path-traced configuration")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` started plainly: N child processes, one per GPU, with the torch.distributed.run
    environment. The parent makes no GPU call (no torch.cuda.*, no rtx_*): it only starts the ranks, relays rank 0's
    JSON line and reports failure if any rank fails."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    out0 = procs[0].communicate()[0]
    codes = [p.wait() for p in procs]
    lines = [ln for ln in out0.splitlines() if ln.startswith('{"metric"')]
    for ln in out0.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if any(codes) or len(lines) != 1:
        # the RCCL ranks did not deliver: the same measurement through the one-process group driver (a fresh child)
        why = f"RCCL ranks exit codes {codes}, {len(lines)} result lines"
        print(f"bench.py: {why}; running the rtx_group driver instead", file=sys.stderr)
        return spawn_group_child(a, why)
    print(lines[0], flush=True)


def spawn_group_child(a, fallback_from=None):
    """One fresh child process that drives all N GPUs through rtx_group_* (group_main). This process makes no GPU call."""
    import subprocess
    env = dict(os.environ, RTX_BENCH_GROUP_CHILD="1", RTX_AMD_NO_TORCH_PRELOAD="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if fallback_from:
        env["RTX_BENCH_FALLBACK_FROM"] = fallback_from
    argv = [x for x in sys.argv[1:] if x != "--group"] + ["--group"]
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    if p.returncode or len(lines) != 1:
        print(f"bench.py: group child exit code {p.returncode}, {len(lines)} result lines", file=sys.stderr)
        sys.exit(p.returncode or 1)
    print(lines[0], flush=True)


def group_main(a):
    """--group: N GPUs, one process, no torch: rtx_group_create / rtx_group_render (csrc/rtx_group.cpp) - one host thread per
    member, scene replicated, buckets b % N, every peer's film copied to the first GPU over its own xGMI link (hipMemcpyPeer) and
    summed there by one kernel (mergeToImage, RayCaster.cpp:1182-1212). The timed region is K calls of rtx_group_render, which
    returns when the merged frame is resolved into host arrays (50 MB over PCIe at 1080p: inside the time, unlike the rank mode)."""
    import raytracer_amd as ra
    from raytracer_amd import synth, _ffi, api
    from raytracer_amd.scene import render_params
    devices = [int(x) for x in a.group_devices.split(",")] if a.group_devices else list(range(a.gpus))
    if len(devices) != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but {len(devices)} group devices", file=sys.stderr)
        sys.exit(2)
    sd = synth.build_scene(a.tri_scale)
    cam = synth.cerealbar_camera(a.width, a.height)
    mode = _ffi.SHADING_LEGACY if a.legacy else _ffi.SHADING_PATH
    # algorithmic bytes per extend ray: an instrumented 2 spp pass on one replica (as in the rank mode)
    ra.set_device(devices[0])
    scene = ra.Scene(sd)
    info = scene.info()
    scene.set_counters(True)
    stc = scene.render(cam, render_params(spp_min=2, max_bounces=a.max_bounces, shading_mode=mode))["stats"]
    scene.set_counters(False)
    scene.close()
    nodes_per_ray = stc["nodes_visited"] / max(1, stc["rays_extend"])
    tris_per_ray = stc["tris_tested"] / max(1, stc["rays_extend"])
    bytes_per_ray = 32 + 16 + nodes_per_ray * info["node_bytes"] + tris_per_ray * info["tri_bytes"]
    g = api.Group(sd, devices)
    rp = render_params(spp_min=a.spp, max_bounces=a.max_bounces, samples_per_pass=a.samples_per_pass, shading_mode=mode)
    for _ in range(a.warmup):
        g.render(cam, rp)
    agg = dict(rays=0, ext_rays=0, extend_ms=0.0, shadow_ms=0.0, shade_ms=0.0, nee_ms=0.0, launches=0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = g.render(cam, rp)
        st = r["stats"]
        agg["rays"] += st["rays_primary"] + st["rays_bounce"] + st["rays_shadow"]
        agg["ext_rays"] += st["rays_extend"]
        for k in ("extend_ms", "shadow_ms", "shade_ms", "nee_ms"):
            agg[k] += st[k]          # the slowest member's kernel time of each frame
        agg["launches"] += st["extend_launches"]
    dt = time.perf_counter() - t0
    n = len(devices)
    # extend rays per member = total / n; the members run side by side, so the frame's extend time is one member's
    ext = roofline_entry("extend_kernel", "ray", agg["ext_rays"] / n, bytes_per_ray, agg["extend_ms"], a.steps, None)
    ext.update({"bytes_per_ray": bytes_per_ray, "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray, "launches": agg["launches"],
                "avg_launch_ms": agg["extend_ms"] / max(1.0, agg["launches"] / n), "per": "member (slowest member's HIP-event time)",
                "traffic": None})
    import numpy as np
    out = {
        "metric": "Mrays/s, cerealBar 1080p 128spp path traced (all ray segments: primary + bounce + shadow hops)",
        "value": agg["rays"] / dt / 1e6, "unit": "Mrays/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "s_per_frame": dt / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"legacy shading mode: cerealbar-synth {a.width}x{a.height} {a.spp} spp, Whitted recursion depth 5" if a.legacy else
                                f"C3: cerealbar-synth {a.width}x{a.height} {a.spp} spp path tracing max_bounces={a.max_bounces}"),
                   "triangles": info["triangles"], "bvh_nodes": info["nodes"], "tri_scale": a.tri_scale,
                   "parallelism": f"rtx_group x{n} (one process, one host thread per GPU, devices {devices}), scene replicated, "
                                  "hipMemcpyPeer merge on the first GPU; frame resolved to host arrays inside the timed region"},
        "rays_per_frame": agg["rays"] / a.steps, "image_finite": bool(np.isfinite(r["linear"]).all()), "mean_linear": float(r["linear"].mean()),
        "roofline": dict(ext, bound="hbm"),
        "kernel_ms_per_frame_slowest_member": {k[:-3]: agg[k] / a.steps for k in ("extend_ms", "shadow_ms", "shade_ms", "nee_ms")},
    }
    if os.environ.get("RTX_BENCH_FALLBACK_FROM"):
        out["fallback_from"] = os.environ["RTX_BENCH_FALLBACK_FROM"]
    g.close()
    print(json.dumps(out), flush=True)


def make_film(W, H, dev):
    import torch
    from raytracer_amd import _ffi
    n = W * H
    t = dict(accum=torch.empty((n, 4), dtype=torch.float32, device=dev), count=torch.empty(n, dtype=torch.int32, device=dev),
             z=torch.empty(n, dtype=torch.float32, device=dev), albedo=torch.empty((n, 4), dtype=torch.float32, device=dev),
             normal=torch.empty((n, 4), dtype=torch.float32, device=dev))
    f = _ffi.Film(t["accum"].data_ptr(), t["count"].data_ptr(), t["z"].data_ptr(), t["albedo"].data_ptr(), t["normal"].data_ptr())
    return t, f


def main():
    a = parse()
    if a.group:
        # the measuring process is always a fresh child of a parent that makes no GPU call
        return group_main(a) if os.environ.get("RTX_BENCH_GROUP_CHILD") else spawn_group_child(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a)
    try:
        return rank_main(a)
    except Exception as e:   # under torch.distributed.run: RCCL / the second HIP runtime failed on this node
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1:
            raise
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            # rank 0 answers for the job through the one-process group driver (a child process: nothing is re-executed here)
            return spawn_group_child(a, f"rank 0 of {world} failed: {type(e).__name__}: {str(e)[:200]}")
        sys.exit(0)   # the other ranks leave quietly: rank 0's line is the job's result


def rank_main(a):
    import numpy as np
    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import synth, _ffi
    from raytracer_amd.scene import render_params
    from raytracer_amd.distributed import reduce_film
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # RTX_BENCH_BACKEND=gloo is for rehearsing the N > 1 code path on a box with fewer GPUs than ranks (ranks then share
    # devices and the film reduce is staged through host memory); the measured configuration is nccl = RCCL, one GPU per rank
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ra.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    sd = synth.build_scene(a.tri_scale)
    cam = synth.cerealbar_camera(a.width, a.height)
    scene = ra.Scene(sd)
    info = scene.info()
    W, H = a.width, a.height
    tens, film = make_film(W, H, dev)
    mode = _ffi.SHADING_LEGACY if a.legacy else _ffi.SHADING_PATH
    rp = render_params(spp_min=a.spp, max_bounces=a.max_bounces, rank=rank, world=world, samples_per_pass=a.samples_per_pass,
                       shading_mode=mode)

    def step():
        ra.film_clear(film, W, H)
        st = scene.render_accumulate(cam, rp, film)
        reduce_film(tens["accum"], tens["count"], tens["z"], dst=0)   # RCCL over xGMI; no-op at world == 1
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # algorithmic bytes per extend ray from an instrumented (counting) pass at 2 spp, untimed
    scene.set_counters(True)
    rp_c = render_params(spp_min=2, max_bounces=a.max_bounces, rank=rank, world=world, shading_mode=mode)
    ra.film_clear(film, W, H)
    stc = scene.render_accumulate(cam, rp_c, film)
    scene.set_counters(False)
    # rays traced by extend_kernel launches (not those of the tail / legacy kernels, which extend_ms does not time)
    ext_rays_c = stc["rays_extend"]
    nodes_per_ray = stc["nodes_visited"] / max(1, ext_rays_c)
    tris_per_ray = stc["tris_tested"] / max(1, ext_rays_c)
    bytes_per_ray = 32 + 16 + nodes_per_ray * info["node_bytes"] + tris_per_ray * info["tri_bytes"]

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    agg = dict(rays=0, ext_rays=0, extend_ms=0.0, shadow_ms=0.0, shade_ms=0.0, nee_ms=0.0, launches=0, shadow_rays=0, vertices=0, shadow_segments=0)
    for _ in range(a.steps):
        st = step()
        agg["rays"] += st["rays_primary"] + st["rays_bounce"] + st["rays_shadow"]
        agg["ext_rays"] += st["rays_extend"]
        agg["shadow_rays"] += st["rays_shadow"]
        agg["extend_ms"] += st["extend_ms"]
        agg["shadow_ms"] += st["shadow_ms"]
        agg["shade_ms"] += st["shade_ms"]
        agg["nee_ms"] += st["nee_ms"]
        agg["launches"] += st["extend_launches"]
        agg["vertices"] += st["vertices"]
        agg["shadow_segments"] += st["shadow_segments"]
    sync()
    dt = time.perf_counter() - t0

    sdev = dev if backend == "nccl" else torch.device("cpu")
    tt = torch.tensor([dt], dtype=torch.float64, device=sdev)
    rays = torch.tensor([float(agg["rays"])], dtype=torch.float64, device=sdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt_max, rays_all = float(tt.item()), float(rays.item())

    if rank == 0:
        res = ra.film_resolve(film, cam)
        finite = bool(np.isfinite(res["linear"]).all())
        default_cfg = (not a.legacy and a.tri_scale == 1.0 and a.width == 1920 and a.height == 1080 and a.spp == 128 and a.max_bounces == 64 and
                       world == 1)
        meas = Measured(default_cfg)
        # ---- roofline of the kernels that make up the frame (algorithmic bytes per unit of work) ----
        frames = a.steps
        kernels = {}
        ext = roofline_entry("extend_kernel", "ray", agg["ext_rays"], bytes_per_ray, agg["extend_ms"], frames,
                             meas.traffic("extend_kernel<false"))
        ext.update({"algorithmic_bytes_per_launch": bytes_per_ray * agg["ext_rays"] / max(1, agg["launches"]),
                    "bytes_per_ray": bytes_per_ray, "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray,
                    "avg_launch_ms": agg["extend_ms"] / max(1, agg["launches"]), "launches": agg["launches"],
                    "extend_Mrays_s": agg["ext_rays"] / max(1e-9, agg["extend_ms"]) / 1e3,
                    "traffic": meas.traffic_per_launch("extend_kernel<false"), "traffic_source": meas.traffic_src,
                    "limiter": meas.limiter("extend_kernel<false")})
        kernels["extend"] = ext
        if not a.legacy:
            # shade: 340 B per path vertex (2 x 64 B path state, 3 x 32 B vertex attributes, 64 B material, <= 3 x 16 B texture taps);
            # since round 3 the light loop's lobe evaluation runs in nee_kernel: the two launches together are the "shade" of 8(d)
            sh = roofline_entry("shade_kernel + nee_kernel", "path vertex", agg["vertices"], 340.0, agg["shade_ms"], frames,
                                meas.traffic("shade_kernel<", "nee_kernel<"))
            sh["nee_ms_per_frame"] = agg["nee_ms"] / frames
            sh["limiter"] = meas.limiter("shade_kernel<")
            kernels["shade"] = sh
            hops_c = max(1, stc["rays_shadow"])
            bytes_per_hop = 32 + 16 + stc["shadow_nodes"] / hops_c * info["node_bytes"] + stc["shadow_tris"] / hops_c * info["tri_bytes"]
            sw = roofline_entry("shadow_kernel", "shadow hop", agg["shadow_rays"], bytes_per_hop, agg["shadow_ms"], frames,
                                meas.traffic("shadow_kernel<false, false, false"))
            sw.update({"nodes_per_hop": stc["shadow_nodes"] / hops_c, "tris_per_hop": stc["shadow_tris"] / hops_c,
                       "segments_per_frame": agg["shadow_segments"] / frames, "limiter": meas.limiter("shadow_kernel<false, false, false")})
            kernels["shadow"] = sw
        out = {
            "metric": "Mrays/s, cerealBar 1080p 128spp path traced (all ray segments: primary + bounce + shadow hops)",
            "value": rays_all / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3,
            "s_per_frame": dt_max / a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"legacy shading mode: cerealbar-synth {W}x{H} {a.spp} spp, Whitted recursion depth 5" if a.legacy else
                                    f"C3: cerealbar-synth {W}x{H} {a.spp} spp path tracing max_bounces={a.max_bounces}"),
                       "triangles": info["triangles"], "bvh_nodes": info["nodes"], "tri_scale": a.tri_scale,
                       "parallelism": f"tile-sharded x{world}, scene replicated, film reduce over " + ("RCCL" if backend == "nccl" else backend + " (rehearsal)")},
            "rays_per_frame": rays_all / a.steps,
            "image_finite": finite,
            "mean_linear": float(res["linear"].mean()),
            # the dominant kernel. `bound` names the roofline its algorithmic bytes are priced against (HBM: nothing on this path
            # is a contraction); what holds it below that line is in `limiter` (PMC: divergence and memory latency)
            "roofline": dict(ext, bound="hbm"),
            "roofline_kernels": kernels,
            "library_sha256": meas.sha,
            "kernel_ms_per_frame_rank0": {"extend": agg["extend_ms"] / a.steps, "shadow": agg["shadow_ms"] / a.steps,
                                          "shade": agg["shade_ms"] / a.steps, "nee": agg["nee_ms"] / a.steps},
        }
        if world == 1 and not a.no_cpu_baseline and a.legacy:
            out["cpu_baseline"], _ = cpu_baseline(sd, cam, a)
        elif world == 1 and not a.no_cpu_baseline:
            # The timed frames above ran with the reference's box culls checked (the default: every sample is the oracle's,
            # DESIGN.md 4 item 8). One frame without the checks (RTX_AMD_BOX_CULLS=0, the lean kernels alone) beside it: what the
            # checks cost, and what leaving them out does to the image.
            os.environ["RTX_AMD_BOX_CULLS"] = "0"
            step()
            sync()
            t1 = time.perf_counter()
            step()
            sync()
            lean_ms = (time.perf_counter() - t1) * 1e3
            del os.environ["RTX_AMD_BOX_CULLS"]
            res_lean = ra.film_resolve(film, cam)
            out["without_box_cull_checks"] = {"switch": "RTX_AMD_BOX_CULLS=0", "ms_per_step": lean_ms, "steps": 1}
            out["cpu_baseline"], cpu_img = cpu_baseline(sd, cam, a)
            if a.cpu_spp in (0, a.spp):   # equal spp: the same samples on both sides
                out["rmse_vs_cpu"] = rmse_vs_cpu(cpu_img, res["linear"], res_lean["linear"], W, H, a)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline_entry(kernel, unit, units, bytes_per_unit, ms, frames, traffic_bytes):
    """achieved = algorithmic bytes / HIP-event time of the kernel's launches (events recorded by the library on the launch
    stream); traffic_bytes: HBM-side bytes of the same launches over one frame from the committed PMC passes, or None."""
    achieved = units * bytes_per_unit / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    e = {"kernel": kernel, "bound": "hbm", "unit": "GB/s", "achieved": achieved, "peak": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS,
         "work_unit": unit, "units_per_frame": units / max(1, frames), "algorithmic_bytes_per_unit": bytes_per_unit, "ms_per_frame": ms / max(1, frames)}
    if traffic_bytes is not None:
        e["traffic_bytes_per_unit"] = traffic_bytes / max(1.0, units / max(1, frames))
    return e


class Measured:
    """Counter-side figures of this workload from the committed PMC passes (tools/pmc_traffic.sh, tools/pmc.sh: a PMC run cannot be
    nested inside this process). They describe the kernels of ONE build: each file records the SHA-256 of the librtx_amd.so it
    was measured on, and a figure is reported only when this run loads that same library and is the configuration it was
    measured on - stale counters do not ride along."""

    def __init__(self, default_cfg):
        import hashlib
        from raytracer_amd import _ffi
        with open(_ffi.LIB_PATH, "rb") as f:
            self.sha = hashlib.sha256(f.read()).hexdigest()
        self.traffic_src = self.limiter_src = None
        self.t = self.l = None
        if not default_cfg:
            return
        for rel in ("profiles/r03/kernel_traffic_c3.json",):
            path = os.path.join(ROOT, rel)
            if os.path.exists(path):
                j = json.load(open(path))
                if j.get("library_sha256") == self.sha:
                    self.t, self.traffic_src = j["kernels"], rel
        for rel in ("profiles/r03/kernel_limiters.json",):
            path = os.path.join(ROOT, rel)
            if os.path.exists(path):
                j = json.load(open(path))
                if j.get("library_sha256") == self.sha:
                    self.l, self.limiter_src = j["kernels"], rel

    def _sum(self, prefixes):
        if not self.t:
            return None, 0
        ks = [v for n, v in self.t.items() if any(n.startswith(p) for p in prefixes)]
        if not ks:
            return None, 0
        # 2 x FETCH_SIZE: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md "HBM")
        # (the instantiations a prefix names - by-segment and front walk, plain and REPACK - are launches of the same kernel)
        return sum((2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0 for k in ks), sum(k["dispatches"] for k in ks)

    def traffic(self, *prefixes):          # bytes per frame (the PMC pass runs one frame)
        return self._sum(prefixes)[0]

    def traffic_per_launch(self, *prefixes):
        b, n = self._sum(prefixes)
        return b / n if b is not None and n else None

    def limiter(self, prefix):
        if not self.l:
            return None
        # the instantiation that carries the full-size frame: the by-segment walk / the bounce without compaction (last template
        # argument false); the counter pass runs a small frame, whose launches mostly take the front walk
        ks = [(n, v) for n, v in self.l.items() if n.startswith(prefix)]
        main = [v for n, v in ks if n.endswith("false>")]
        k = main[0] if main else (max((v for _, v in ks), key=lambda v: v.get("waves", 0.0)) if ks else None)
        return dict(k, source=self.limiter_src) if k else None


def rmse_vs_cpu(cpu_img, gpu_default, gpu_lean, W, H, a):
    """The GPU film against the CPU render of the cpu_baseline leg, on the pixels both hold completely: the 14 x 14 interior
    pixels of the sampled 16 x 16 buckets (tent filter, radius 1: an interior pixel receives samples of its own bucket only)."""
    import numpy as np
    m = np.zeros((H, W), bool)
    for by in range(0, H, 16 * 8):
        for bx in range(0, W, 16 * 8):
            m[by + 1:min(by + 15, H - 1), bx + 1:min(bx + 15, W - 1)] = True
    c = cpu_img.reshape(H, W, 3)[m].astype(np.float64)
    mean_lum = float(c.mean())

    def rel(g):
        d = g.reshape(H, W, 3)[m].astype(np.float64) - c
        return float(np.sqrt((d * d).mean()) / max(mean_lum, 1e-30))

    return {"pixels": int(m.sum()), "spp": a.cpu_spp if a.cpu_spp > 0 else a.spp, "mean_luminance": mean_lum,
            "rmse_over_mean": rel(gpu_default), "bar": 1e-5, "rmse_over_mean_without_box_cull_checks": rel(gpu_lean),
            "note": "per-sample radiance is bit-identical to the CPU render; the films differ by the order of their float sums. Without "
                    "the box-cull checks (RTX_AMD_BOX_CULLS=0) about 2 samples in 10^7 end differently (DESIGN.md 4 item 8)"}


def cpu_baseline(sd, cam, a):
    from raytracer_amd import _ffi
    from raytracer_amd.scene import render_params
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded sample of the
    same workload: every 8th 16x16 bucket in x and y at a reduced sample count."""
    from tests.oracle_ffi import OracleScene
    # every host core this job may use: the GPU box exports its CPU share as OMP_NUM_THREADS (os.cpu_count() there is the
    # whole machine, most of which belongs to other jobs)
    threads = int(os.environ.get("OMP_NUM_THREADS") or 0) or (os.cpu_count() or 1)
    cpu_spp = a.cpu_spp if a.cpu_spp > 0 else a.spp
    t0 = time.perf_counter()
    osc = OracleScene(sd)
    build_s = time.perf_counter() - t0
    rp = render_params(spp_min=cpu_spp, max_bounces=a.max_bounces,
                       shading_mode=_ffi.SHADING_LEGACY if a.legacy else _ffi.SHADING_PATH)
    r = osc.render(cam, rp, nthreads=threads, subset_stride=8)
    rays = r["rays"]["primary"] + r["rays"]["bounce"] + r["rays"]["shadow"]
    return ({"value": rays / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
             "sample": f"every 8th 16x16 bucket in x and y (1/64 of the frame) at {cpu_spp} spp, same scene/camera/sampler; "
                       f"{rays} ray segments in {r['seconds']:.2f} s (+{build_s:.1f} s BVH build, untimed)",
             "seconds": r["seconds"], "paths": r["rays"]["primary"]}, r["linear"])


if __name__ == "__main__":
    main()
