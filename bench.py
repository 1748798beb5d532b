#!/usr/bin/env python3
"""Headline benchmark: Mrays/s, primary closest-hit rays on the 1M-triangle scene.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or bare:
     without WORLD_SIZE in the environment bench.py starts those N ranks itself as a child process and relays rank 0's line;
     --gpus N that disagrees with WORLD_SIZE is an error, not a 1-GPU line)

A "step" is one pass of the hot path over one batch: every rank traces its own 4096x4096
frame (2^24 rays; N=1 is BASELINE.json configs[1], N>1 is configs[3]: frame r on rank r,
BVH replicated) and, for N>1, the 16-byte hit records are gathered onto rank 0 over RCCL.
Rays, BVH and hit records are resident in HBM for the whole timed region. The BVH is
built on the GPU by the product path (rtk_dev_scene_build) before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job Mrays/s, plus
  roofline:     algorithmic bytes of the traversal kernel per unit it actually fetches for (a 64-ray tile for
                the packet kernel, a ray for the per-lane kernels) / its measured duration vs 8 TB/s, the
                fabric traffic of the same launch from the committed rocprofv3 PMC summary, and the kernel's
                real limiter (VALU issue / lane utilisation) from the same counters
  cpu_baseline: the CPU oracle (a port of rtk.c's trace path) timed on this host's cores,
                with the parity of the timed GPU result against it (every id mismatch is listed).

Other BASELINE.json configs (parity-test cases, not the headline line):
  --workload incoherent   config 3: 2^24 random rays on the same scene
  --workload shadow       config 5: 10M-triangle scene, GPU build timed, 2^24 any-hit shadow rays
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 128, 48, 32, 16
RAY_DTYPE_NP = np.dtype([("origin", "<f4", (3,)), ("direction", "<f4", (3,)), ("min_t", "<f4"), ("max_t", "<f4")])
HIT_RECORD_NP = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])


def log(*a):
    print(*a, file=sys.stderr, flush=True)


CLOCK_GHZ = 2.4                 # MI355X max engine clock (MI355X_MICROARCH.md); used only when a summary lacks SQ_BUSY_CYCLES
SIMDS = 256 * 4
SHADER_ENGINES = 32             # 8 XCDs x 4: SQ_BUSY_CYCLES is summed over them


def kernel_code_sha16(workload):
    """Identity of the workload's traversal kernel(s): a hash over their gfx950 machine code inside the built library
    (rtk_amd/kernel_id.py). Comments, host code and other kernels do not change it."""
    from rtk_amd import api, kernel_id
    return kernel_id.workload_kernel_sha16(api.LIB_PATH, workload)


def load_pmc_summary(workload):
    """Newest committed rocprofv3 PMC summary of this workload (profiles/rNN_<workload>_lbvh_pmc.json), and whether it
    was measured on the traversal kernels as they are now (scripts/summarize_profile.py records the hash of their code)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_lbvh_pmc.json" % workload)), reverse=True):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if "hbm_traffic_bytes_per_launch" not in pj:
            continue
        mine = kernel_code_sha16(workload)
        return pj, os.path.relpath(f, ROOT), (mine is not None and pj.get("kernel_code_sha16") == mine)
    return None, None, False


L2_MISS_CEILING_G_PER_S = 56.0


def limiter_from_pmc(pj):
    """What the counters say bounds the kernel: share of cycles the VALU pipes are issuing, lane utilisation of
    those instructions, share of wave time spent waiting, L2 hit rate."""
    c = pj.get("pmc_per_launch", {})
    ns = pj.get("kernel_trace", {}).get("average_ns")

    def m(k):
        return c[k]["mean"] if k in c else None
    out = {}
    # the clock the kernel actually ran at under the profiler: busy cycles summed over the shader engines / its duration
    clock = m("SQ_BUSY_CYCLES") / SHADER_ENGINES / ns if (m("SQ_BUSY_CYCLES") and ns) else CLOCK_GHZ
    out["clock_ghz_measured"] = round(clock, 3)
    if m("SQ_ACTIVE_INST_VALU") and ns:
        # SQ_ACTIVE_INST_* count quad-cycles summed over all SIMDs (MI355X_MICROARCH.md, cycle constants)
        out["valu_busy"] = round(m("SQ_ACTIVE_INST_VALU") * 4.0 / (SIMDS * ns * clock), 3)
    if m("SQ_INSTS_SALU") and ns:
        # one scalar ALU per CU, shared by its four SIMDs, one instruction per cycle (scalar memory instructions and
        # branches go through the same issue port)
        scalar = m("SQ_INSTS_SALU") + (m("SQ_INSTS_SMEM") or 0.0) + (m("SQ_INSTS_BRANCH") or 0.0)
        out["salu_issue"] = round(scalar / (SIMDS / 4.0 * ns * clock), 3)
        out["instructions_per_64_rays"] = {k[9:].lower(): round(m(k) * 64.0 / pj["rays_per_launch"], 1) for k in
                                           ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS")
                                           if m(k) and pj.get("rays_per_launch")}
    if m("SQ_THREAD_CYCLES_VALU") and m("SQ_ACTIVE_INST_VALU"):
        out["valu_lane_utilisation"] = round(m("SQ_THREAD_CYCLES_VALU") / (64.0 * m("SQ_ACTIVE_INST_VALU")), 3)
    if m("SQ_WAIT_ANY") and m("SQ_WAVE_CYCLES"):
        out["wave_cycles_waiting"] = round(m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), 3)
    if "l2_hit_rate" in pj:
        out["l2_hit_rate"] = round(pj["l2_hit_rate"], 3)
    if m("TCC_MISS_sum") and ns:
        # L2 misses per second against what the memory system gave a pure gather kernel (scripts/gather_probe.hip,
        # profiles/r03_gather_probe.log: 56 G requests/s whether 16, 64 or 128 bytes of the line are used, from the Infinity
        # Cache and from HBM alike, at 4 to 32 waves per CU)
        out["l2_miss_g_per_s"] = round(m("TCC_MISS_sum") / ns, 1)
        out["l2_miss_ceiling_g_per_s"] = L2_MISS_CEILING_G_PER_S
    return out


FRAC_NOTE = ("the ALGORITHMIC bytes (every visit priced at its record size, SURVEY.md 8d) divided by the kernel's time exceed the 8 TB/s "
             "HBM peak: the caches serve most of them (see limiter.l2_hit_rate), so that quotient is cache demand and NOT a fraction of "
             "the HBM roofline -- `frac` is null; `real_bound` names what the counters say binds the kernel, `frac_by_traffic` is "
             "the measured fabric traffic over the same time")


def honest_fractions(achieved_gbs, traffic, k_ms, lim):
    """(frac, frac_note, frac_by_traffic, real_bound). `frac` is achieved / peak only while that is a statement about HBM
    (<= 1); `real_bound` lists every ceiling the committed counters can price -- fabric traffic against the HBM peak, VALU
    issue slots (busy share of the four SIMDs' pipes; times the lane utilisation = useful lane-operations against the chip's
    VALU peak), the CU's scalar issue port, L2 misses per second against the 56 G/s the memory system gave a pure gather
    kernel -- and names the largest."""
    over = achieved_gbs > HBM_PEAK_GBS
    frac = None if over else round(achieved_gbs / HBM_PEAK_GBS, 4)
    fbt = round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None
    real = None
    if lim:
        c = {}
        if fbt is not None:
            c["hbm_traffic_of_8_TB_s"] = fbt
        if lim.get("valu_busy") is not None:
            c["valu_issue_slots"] = lim["valu_busy"]
            if lim.get("valu_lane_utilisation") is not None:
                c["valu_lane_ops_of_peak"] = round(lim["valu_busy"] * lim["valu_lane_utilisation"], 3)
        if lim.get("salu_issue") is not None:
            c["scalar_issue_port"] = lim["salu_issue"]
        if lim.get("l2_miss_g_per_s") is not None:
            c["l2_misses_of_56_G_s"] = round(lim["l2_miss_g_per_s"] / L2_MISS_CEILING_G_PER_S, 3)
        binding = [k for k in c if k != "valu_lane_ops_of_peak"]
        if binding:
            top = max(binding, key=lambda k: c[k])
            real = dict(c, binding=top, binding_frac=c[top])
    return frac, (FRAC_NOTE if over else None), fbt, real


def f32_ulps(a, b):
    return int(abs(int(np.float32(a).view(np.int32)) - int(np.float32(b).view(np.int32))))


def other_workload(kind, steps, warmup, frame=4096, with_parity=True):
    """Compact line for one of the other BASELINE.json configs (config 3: incoherent rays on the 1M-triangle scene; config 5:
    10M-triangle GPU build + any-hit shadow rays), measured in the same process with the library's defaults: value, kernel
    time from events on the launch stream, the per-ray algorithmic bytes of the counting build, roofline fraction, and the
    committed counter summary's traffic if it was taken on these kernels. No CPU leg (the parity tests cover these)."""
    import torch
    from rtk_amd import api, synth
    shadow = kind == "shadow"
    cfg = synth.CONFIGS[5 if shadow else 3]
    n = frame * frame
    t0 = time.time()
    # (scene and rays are made on the GPU: the same counter-based generator, bit for bit -- tests/test_cabi.py)
    d_tris = synth.t_triangle_soup(cfg["num_tris"], cfg["spread"], cfg["scene_seed"])
    torch.cuda.synchronize()
    ds = api.DeviceScene.build([dict(positions=d_tris)])
    ds.free()
    ds = api.DeviceScene.build([dict(positions=d_tris)])          # second build: steady state (the first pays code-object load)
    build_ms = ds.info()["build_ms"]
    d_rays = synth.t_rays_shadow(n) if shadow else synth.t_rays_incoherent(n)
    rays = d_rays
    opts = api.make_opts(sort_rays=True)      # both per-lane workloads: re-ordered by entry cell inside every timed step (3.0 -> 3.5 / 2.5 -> 4.35 Grays/s)
    out_bytes = 1 if shadow else HIT_BYTES
    d_out = torch.empty(n * out_bytes, dtype=torch.uint8, device="cuda")
    trace = (lambda: ds.trace_any_device(d_rays, n, d_out, opts)) if shadow else (lambda: ds.trace_device(d_rays, n, d_out, opts))
    _, ctr = ds.trace_any_counted(rays, opts) if shadow else ds.trace_counted(rays, opts)
    lane_node_bytes = NODE_BYTES if os.environ.get("RTK_AMD_QNODES", "1") == "0" else 64
    alg_bytes = n * (RAY_BYTES + out_bytes) + ctr["nodes"] * lane_node_bytes + ctr["triangles"] * TRI_BYTES
    alg_bytes += n * (32 + 8 + 3 * 24 + 8)              # the re-ordering pre-pass inside the step (see main): keys, three radix passes, the word read back
    for _ in range(warmup):
        trace()
    torch.cuda.synchronize()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    ev_a.record()
    for _ in range(steps):
        trace()
    ev_b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    k_ms = ev_a.elapsed_time(ev_b) / steps      # one event pair around the K steps: the step's time on the GPU
    res = d_out.cpu().numpy()
    hit_frac = float(res.astype(bool).mean()) if shadow else float((res.view(np.uint32).reshape(-1, 4)[:, 3] != 0xFFFFFFFF).mean())
    # ---- sampled parity of the TIMED buffer against the oracle on the same BVH (the blob exported from this scene): 2^16 rays
    parity = None
    if with_parity:
        try:
            from oracle import pyoracle
            tp = time.time()
            sel = np.arange(0, n, max(1, n >> 16))
            d_sel = torch.from_numpy(sel).cuda()
            rays_h = np.ascontiguousarray(d_rays[d_sel].cpu().numpy()).view(RAY_DTYPE_NP).reshape(-1)
            exported = pyoracle.Blob(ds.export_blob())
            out_s = pyoracle.SpreadBuffer(len(sel) * 16)
            orec = pyoracle.trace_records(exported, rays_h, out_s).copy()
            out_s.free()
            om = orec["triangle_index"] != 0xFFFFFFFF
            if shadow:
                # (any-hit parity = the flag equals the oracle's closest-hit boolean)
                parity = {"rays": int(len(sel)), "occluded_flag_mismatches": int((res[sel].astype(bool) != om).sum())}
            else:
                g = res.view(HIT_RECORD_NP)[sel]
                gm = g["prim"] != 0xFFFFFFFF
                both = gm & om
                same = both & (g["prim"] == orec["triangle_index"])
                parity = {"rays": int(len(sel)), "hit_miss_mismatches": int((gm != om).sum()), "id_mismatches": int((both & ~same).sum()),
                          "tuv_bit_exact_fraction": float(np.mean((g["t"][same].view(np.uint32) == orec["t"][same].view(np.uint32)) &
                                                                  (g["u"][same].view(np.uint32) == orec["u"][same].view(np.uint32)) &
                                                                  (g["v"][same].view(np.uint32) == orec["v"][same].view(np.uint32)))) if same.any() else 1.0}
            parity["against"] = "the CPU oracle traversing the blob exported from this GPU-built scene, every %d-th ray of the timed batch" % max(1, n >> 16)
            parity["seconds"] = round(time.time() - tp, 2)
            del exported
        except Exception as e:
            parity = {"error": repr(e)}
    # ---- config 5 end to end: build + re-order + trace as ONE wall-clock interval
    end_to_end = None
    if shadow:
        def once(positions):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ds2 = api.DeviceScene.build([dict(positions=positions)])
            ds2.trace_any_device(d_rays, n, d_out, opts)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
            b = ds2.info()["build_ms"]
            ds2.free()
            return dt, b
        ds.free()
        ds = None
        e_dev = min(once(d_tris) for _ in range(3))
        host_tris = d_tris.cpu().numpy()
        e_host = min(once(host_tris) for _ in range(2))
        del host_tris
        end_to_end = {"from_device_resident_vertices_ms": round(e_dev[0], 3), "of_which_build_ms": round(e_dev[1], 3),
                      "from_host_memory_ms": round(e_host[0], 3), "of_which_build_incl_pcie_ms": round(e_host[1], 3),
                      "what": "wall clock around rtk_dev_scene_build + the re-ordering pre-pass + the any-hit trace of all %d rays (rays resident in HBM), best of 3 / 2" % n}
    pj, src, fresh = load_pmc_summary(kind)
    traffic = float(pj["hbm_traffic_bytes_per_launch"]) if (pj and fresh) else None
    lim = limiter_from_pmc(pj) if (pj and fresh) else None
    if ds is not None:
        ds.free()
    del d_rays, d_out, d_tris
    torch.cuda.empty_cache()
    from rtk_amd import api as _api
    _api.lib().rtk_amd_release_workspace()
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    frac, frac_note, fbt, real_bound = honest_fractions(achieved, traffic, k_ms, lim)
    return {"workload": ("config5: 10M-tri soup, GPU LBVH build + %d any-hit shadow rays (re-ordered by entry cell inside every step)" % n) if shadow
            else ("config3: 1M-tri soup, %d incoherent rays (re-ordered by entry cell inside every step)" % n),
            "value": round(n * steps / elapsed / 1e6, 2), "unit": "Mrays/s", "steps": steps, "kernel_ms": round(k_ms, 4),
            "hit_fraction": round(hit_frac, 4), "bvh_build_ms_device_resident_mesh": round(build_ms, 3),
            "algorithmic_bytes_per_launch": int(alg_bytes), "achieved_gb_s": round(achieved, 1), "frac": frac, "frac_note": frac_note,
            "frac_by_traffic": fbt, "real_bound": real_bound,
            "visits_per_ray": {"nodes": round(ctr["nodes"] / n, 2), "triangles": round(ctr["triangles"] / n, 2)},
            "traffic": traffic, "traffic_source": src if traffic else ("none: %s was measured on other kernel code" % src if pj else None),
            "limiter": {k: v for k, v in lim.items()} if lim else None,
            "ray_reordering_inside_the_step": True, "kernel": ("rtk_lane_hot_any" if shadow else "rtk_lane_hot_closest") + " (hand-written gfx950 assembly) + rtk_trace_kernel on the rays it hands back",
            "parity_vs_oracle_same_bvh": parity, "end_to_end": end_to_end,
            "setup_s": round(time.time() - t0, 1)}


def launch_ranks(n):
    """`bench.py --gpus N` without a launcher around it: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child process on 127.0.0.1 and a free port, pass its stdout (rank 0's one JSON line) and stderr through,
    return its exit code. The parent imports neither torch nor the library: a process that has initialised the GPU must not
    start (let alone exec) the ranks."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's peer buffers (task statement, Environment)
    log("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 timed steps behind 20 warm-up steps (0.12 s in all for the 1 ms frame of the default workload). The clocks of an
    # idle MI355X take ~15 frames to come up (scripts/steps_probe.py: 1.11, 1.08, 1.07, ... 0.99 ms per frame from a cold start), so a
    # handful of steps measures the ramp, not the rate
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="coherent", choices=["coherent", "incoherent", "shadow"])
    ap.add_argument("--bvh", default="device", choices=["device", "oracle-blob", "cpu-sah"],
                    help="device = GPU LBVH build (product path); oracle-blob = upload a blob built by the CPU oracle (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave hit records on their GPU (same as --gather none)")
    ap.add_argument("--gather", default="striped", choices=["striped", "root", "none"],
                    help="N>1, what `value` includes: striped = every rank ends up with its stripe of every shard (all-to-all of record "
                         "slices: no link carries more than 1/N of a shard per step); root = all records onto rank 0 (bound by the root's "
                         "incoming links); none = records stay where they were traced. The other two are timed as well and reported in config.")
    ap.add_argument("--static", action="store_true", help="A/B: one fixed ray per lane instead of persistent refill")
    ap.add_argument("--no-tiling", action="store_true")
    ap.add_argument("--no-packet", action="store_true", help="A/B: image-shaped batch on the per-lane kernel")
    ap.add_argument("--sort-rays", action="store_true", help="reorder the batch by entry cell first (inside the timed step); the default for --workload incoherent and shadow")
    ap.add_argument("--no-sort-rays", action="store_true", help="incoherent / shadow workloads: trace the batch in the order given")
    ap.add_argument("--refill-min", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--node-exit", type=int, default=0)
    ap.add_argument("--frame", type=int, default=4096)
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default run (N=1, coherent): do not append the compact config-3 / config-5 lines (other_workloads)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's main figure (default: 16, a one-GPU box's share)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU: gloo + CPU tensors + a stand-in tracer that only fills buffers; exercises the "
                         "multi-rank step loop / gather / timing / JSON plumbing in the CPU tests. Prints no perf claim.")
    args = ap.parse_args()
    DRY = args.dry_run_cpu
    if args.workload in ("shadow", "incoherent") and not args.no_sort_rays:
        # 2^24 rays in random order are bound by L2 misses (the 56 G requests/s wall); the library's
        # RTK_TRACE_SORT_RAYS pre-pass (entry-cell Morton order, part of every timed step) takes that wall away:
        # shadow 2.5 -> 4.35 Grays/s, incoherent 3.0 -> 3.5 with the assembly kernels
        args.sort_rays = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process becomes the launcher and NEVER touches a GPU (no torch import, no HIP
        # call before or after): it starts the N ranks as a CHILD `python -m torch.distributed.run`, relays rank 0's JSON line and
        # the child's exit code
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    from rtk_amd import api, shard, synth
    from rtk_amd.types import HIT_RECORD_DTYPE

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a line that says n_gpus = WORLD_SIZE under a command that asked for --gpus N would be read as the N-GPU figure
        log("error: --gpus %d but WORLD_SIZE %d: refusing to print a line for another world size" % (args.gpus, world))
        sys.exit(2)
    if not DRY and world > 1 and torch.cuda.device_count() < world:
        log("error: --gpus %d but this node shows %d GPUs" % (world, torch.cuda.device_count()))
        sys.exit(2)
    dev = "cpu" if DRY else "cuda"

    def sync():
        if not DRY:
            torch.cuda.synchronize()

    if world > 1:
        args.no_cpu_baseline = True      # the CPU leg is rank 0's at N = 1 only (torch.distributed.run also pins OMP_NUM_THREADS=1)
    if DRY:
        args.no_cpu_baseline = True
        if world > 1:
            dist.init_process_group(backend="gloo")
    else:
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        api.lib().rtk_amd_set_device(local_rank)
        if world > 1:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    W = H = args.frame
    n = W * H
    shadow = args.workload == "shadow"
    cfg = synth.CONFIGS[5 if shadow else 2]

    # ---- scene: replicated on every rank -------------------------------------------------
    class _StandIn:
        """--dry-run-cpu only: fills the output with a pattern; nothing is traced."""
        def info(self):
            return dict(num_nodes=0, build_ms=0.0)

        def trace_device(self, d_rays, n, buf, opts):
            buf.fill_((rank * 13 + 1) % 251)

        trace_any_device = trace_device

        def trace_counted(self, rays, opts):
            return None, dict(nodes=0, leaves=0, triangles=0, wave_node_steps=0, wave_triangle_steps=0)

        trace_any_counted = trace_counted

        def trace_packet_counted(self, rays, opts):
            return None, None

    t0 = time.time()
    tris = None if DRY else synth.triangle_soup(cfg["num_tris"], cfg["spread"], cfg["scene_seed"])
    t_gen = time.time() - t0
    t0 = time.time()
    oracle_blob = None
    if DRY:
        ds = _StandIn()
        bvh_kind = "none (dry run)"
    elif args.bvh == "device":
        ds = api.DeviceScene.build([dict(positions=tris)])
        bvh_kind = "gpu-lbvh"
    elif args.bvh == "cpu-sah":
        # tree-quality experiment: the product's own task-graph CPU builder (binned SAH, top down) builds
        # the blob, the GPU traces it. RTK_AMD_CPU_SAH_SPLIT_COST=0.5 RTK_AMD_CPU_LEAF_MIN=1 give the one-triangle leaves
        # of the device build, so that only the topology differs.
        import ctypes as C
        from rtk_amd.types import MeshSet, SceneHeader
        L = api.lib()
        ms = MeshSet([dict(positions=tris)])
        L.rtk_amd_set_builder(1)
        p = L.rtk_build_scene(C.byref(ms.desc))
        L.rtk_amd_set_builder(0)
        if not p:
            raise RuntimeError("cpu builder: " + str(api.last_error()))
        hdr = SceneHeader.from_address(p)
        blob_bytes = np.ctypeslib.as_array((C.c_uint8 * hdr.size_in_bytes).from_address(p)).copy()
        L.rtk_free_scene(p)
        ds = api.DeviceScene.upload(blob_bytes)
        bvh_kind = "cpu-task-graph-sah-blob-upload (split cost %s, leaf min %s)" % (os.environ.get("RTK_AMD_CPU_SAH_SPLIT_COST", "1"), os.environ.get("RTK_AMD_CPU_LEAF_MIN", "4"))
    else:
        from oracle import pyoracle
        oracle_blob = pyoracle.build_scene([dict(positions=tris)])
        ds = api.DeviceScene.upload(oracle_blob)
        bvh_kind = "oracle-sah-blob-upload"
    sync()
    t_build = time.time() - t0
    info = ds.info()
    build_ms_first_call = info["build_ms"]
    build_ms_device_mesh = None
    if args.bvh == "device" and not DRY:
        # the first build of a process also pays one-time costs (code-object load, first touch of the
        # host pages by the DMA engine); build again for the steady-state figure, and once more with the
        # mesh already resident in HBM (no PCIe upload inside the build)
        ds.free()
        ds = api.DeviceScene.build([dict(positions=tris)])
        info = ds.info()
        d_tris = torch.from_numpy(tris).cuda()
        sync()
        ds2 = api.DeviceScene.build([dict(positions=d_tris)])
        build_ms_device_mesh = ds2.info()["build_ms"]
        ds2.free()
        del d_tris
    if rank == 0:
        log("scene: %d tris generated in %.2fs, bvh (%s) in %.2fs: %s" % (cfg["num_tris"], t_gen, bvh_kind, t_build, info))

    # ---- rays: frame `rank` of config 4 (frame 0 == config 2) ------------------------------
    common = dict(static=args.static, refill_min=args.refill_min, blocks_per_cu=args.blocks_per_cu, node_exit=args.node_exit,
                  no_packet=args.no_packet, sort_rays=args.sort_rays)
    if args.workload == "coherent":
        rays = synth.rays_pinhole(W, H, jitter=synth.frame_jitter(rank))
        opts = api.make_opts(image=None if args.no_tiling else (W, H), no_detect=args.no_tiling, **common)
        workload = "config2: 1M-tri soup (seed 1, spread 0.02), %dx%d coherent pinhole primary rays" % (W, H)
        metric = "Mrays/sec (primary, closest-hit) on 1M-tri scene"
    elif args.workload == "incoherent":
        rays = synth.rays_incoherent(n, first=rank * n)
        opts = api.make_opts(**common)
        workload = "config3: 1M-tri soup, %d incoherent rays" % n
        metric = "Mrays/sec (incoherent, closest-hit) on 1M-tri scene"
    else:
        rays = synth.rays_shadow(n, first=rank * n)
        opts = api.make_opts(**common)
        workload = "config5: 10M-tri soup (spread 0.01), GPU LBVH build + %d any-hit shadow rays" % n
        metric = "Mrays/sec (any-hit shadow) on 10M-tri scene"
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)) if DRY else api.to_device(rays)
    out_bytes = 1 if shadow else HIT_BYTES
    # two output buffers: with N > 1 the gather of step k overlaps the trace of step k+1
    d_outs = [torch.empty(n * out_bytes, dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    sizes = [n * out_bytes] * world
    if args.no_gather:
        args.gather = "none"
    mode = args.gather if world > 1 else "none"
    # receive buffers of the two exchanges (two each: the exchange of step k overlaps the trace of step k+1)
    gathered = [torch.empty(sum(sizes), dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None for _ in d_outs]
    striped = [torch.empty(n * out_bytes, dtype=torch.uint8, device=dev) if world > 1 else None for _ in d_outs]
    pending = [[] for _ in d_outs]

    def trace(buf):
        if shadow:
            ds.trace_any_device(d_rays, n, buf, opts)
        else:
            ds.trace_device(d_rays, n, buf, opts)

    def step(k, ev=None, how=None):
        how = how or mode
        b = k % len(d_outs)
        shard.gather_records_wait(pending[b])       # this buffer's previous exchange has drained
        pending[b] = []
        if ev:
            ev[0].record()                          # same stream the kernel is launched on
        trace(d_outs[b])
        if ev:
            ev[1].record()
        if how == "root":
            _, pending[b] = shard.gather_records_start(d_outs[b], sizes, dst=0, out=gathered[b])
        elif how == "striped":
            _, pending[b], _ = shard.exchange_striped_start(d_outs[b], out_bytes, out=striped[b], counts=[n] * world)

    def drain():
        for b in range(len(d_outs)):
            shard.gather_records_wait(pending[b])
            pending[b] = []

    # ---- algorithmic bytes from the counting build (not timed) ----------------------------
    _, ctr = ds.trace_any_counted(rays, opts) if shadow else ds.trace_counted(rays, opts)
    packet_kernel = args.workload == "coherent" and not args.no_tiling and not args.no_packet and not DRY
    # SURVEY.md section 8d per unit the kernel actually fetches for: the packet kernel fetches a node / triangle ONCE
    # per 64-ray tile (scalar cache), the per-lane kernels once per ray.
    # the per-lane kernels read the 64-byte compressed nodes (DevNodeQ) unless RTK_AMD_QNODES=0, the packet kernel the 128-byte exact ones
    lane_node_bytes = NODE_BYTES if os.environ.get("RTK_AMD_QNODES", "1") == "0" else 64
    per_ray_bytes = n * (RAY_BYTES + out_bytes) + ctr["nodes"] * (NODE_BYTES if packet_kernel else lane_node_bytes) + ctr["triangles"] * TRI_BYTES
    pk_ctr = None
    if packet_kernel:
        # the counting form of the kernel that is TIMED: rtk_packet_count2 = rtk_packet_beam2.S assembled with -DRTK_COUNT (three scalar
        # counters per pair of tiles), plus the counting build of the C++ packet kernel on the tiles it hands back
        try:
            _, pk_ctr = ds.trace_packet_counted(rays, opts)
        except api.RtkError as e:
            log("packet counters not available (%s): pricing the C++ packet kernel's steps" % e)
    if packet_kernel and pk_ctr:
        fetched_nodes = pk_ctr["node_steps"] + pk_ctr["handed_back_node_steps"]
        fetched_tris = pk_ctr["triangles_fetched"] + pk_ctr["handed_back_triangle_steps"]
        alg_bytes = n * (RAY_BYTES + out_bytes) + fetched_nodes * NODE_BYTES + fetched_tris * TRI_BYTES
        unit = ("PAIR of adjacent 8x8-pixel tiles (128 rays, one wave): 32 B per ray in, 16 B per ray out, each node (128 B) priced once per node "
                "step of the pair and each triangle record (48 B) once per triangle of a leaf the pair enters; steps counted by rtk_packet_count2, "
                "the kernel that is timed assembled with -DRTK_COUNT (rtk_dev_trace_rays_packet_counted), plus the C++ packet kernel's own "
                "counting build on the %d tiles handed back" % pk_ctr["tiles_handed_back"])
    elif packet_kernel:
        alg_bytes = n * (RAY_BYTES + out_bytes) + ctr["wave_node_steps"] * NODE_BYTES + ctr["wave_triangle_steps"] * TRI_BYTES
        unit = ("tile of 64 rays: each node (128 B) and triangle (48 B) priced once per tile; visit and step counts are those of the counting "
                "C++ kernel (per-lane slab tests, one tile per wave)")
    else:
        alg_bytes = per_ray_bytes
        unit = "ray: each lane fetches its own nodes (%d B) and triangles (48 B)" % lane_node_bytes
        if args.sort_rays:
            # the re-ordering pre-pass is inside the timed region: keys (32 B ray read, one 8-B word written: cell key over the
            # ray's number), two radix passes over the words (8 B histogram read + 8 B read + 8 B written each), and the
            # 8-B word the traversal reads per ray instead of counting
            alg_bytes += n * (32 + 8 + 3 * 24 + 8)
            per_ray_bytes = alg_bytes
            unit += "; plus the ray re-ordering pre-pass (120 B per ray: keys, three radix passes over 8-byte words, the word read back), timed with the traversal"
    sync()

    for k in range(args.warmup):
        step(k)
    drain()
    sync()
    if world > 1:
        dist.barrier()
        sync()

    # ---- timed region --------------------------------------------------------------------
    class _WallEvent:
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    mk_event = _WallEvent if DRY else (lambda: torch.cuda.Event(enable_timing=True))
    # ONE pair of events around the K steps, on the launch stream (an event pair per step put two marker packets between
    # consecutive frames: ~10 us of idle GPU per 0.95 ms frame); kernel_ms = their distance / K = the step's time on the GPU
    ev_region = (mk_event(), mk_event())
    t_start = time.perf_counter()
    ev_region[0].record()
    for k in range(args.steps):
        step(k)
    ev_region[1].record()
    drain()
    sync()
    if world > 1:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t_start
    kernel_ms = [ev_region[0].elapsed_time(ev_region[1]) / max(1, args.steps)]

    # ---- the same frame WITHOUT the image hint (opts = NULL, what a host that only knows rays and records passes): the library looks
    # at the batch (two small launches and a wait per call), finds the image and runs the same kernels; records must be the same bytes
    no_hint = None
    any_image = None
    if packet_kernel and world == 1 and args.steps:
        ref_out = d_outs[(args.steps - 1) % len(d_outs)].clone()
        k2 = max(3, args.steps // 4)
        ds.trace_device(d_rays, n, d_outs[0], None)
        sync()
        per_call = []
        for _ in range(k2):
            t2 = time.perf_counter()
            ds.trace_device(d_rays, n, d_outs[0], None)
            per_call.append(time.perf_counter() - t2)
        sync()
        # (every un-hinted call waits for its stream once -- the look's verdict --, so the host's time from call to call IS the step
        # time: the first call of the loop only enqueues behind nothing and is left out)
        dt = sum(per_call[1:])
        k2 -= 1
        no_hint = {"value": round(n * k2 / dt / 1e6, 2), "unit": "Mrays/s", "steps": k2, "ms_per_step": round(dt / k2 * 1e3, 4),
                   "records_identical_to_the_hinted_run": bool(torch.equal(ref_out, d_outs[0])),
                   "what": "rtk_dev_trace_rays(opts = NULL) on the same rays: the image is detected per call (k_detect_row, k_detect_check, one stream wait)"}
        d_outs[(args.steps - 1) % len(d_outs)].copy_(ref_out)
        # ... and the same rays as an ANY-HIT batch with the hint (rtk_dev_trace_rays_any: one flag per ray): rtk_packet_any2, the
        # any-hit form of the same kernel (a ray is retired at its first hit, a pair of tiles ends when every ray has its answer)
        any_image = None
        try:
            d_flags = torch.empty(n, dtype=torch.uint8, device=dev)
            ds.trace_any_device(d_rays, n, d_flags, opts)
            sync()
            t2 = time.perf_counter()
            for _ in range(k2):
                ds.trace_any_device(d_rays, n, d_flags, opts)
            sync()
            dt2 = time.perf_counter() - t2
            want = ref_out.view(torch.int32).view(-1, 4)[:, 3] != -1
            any_image = {"value": round(n * k2 / dt2 / 1e6, 2), "unit": "Mrays/s", "steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 4),
                         "flags_equal_closest_hit_exists": bool(torch.equal(d_flags != 0, want)),
                         "what": "rtk_dev_trace_rays_any with the image hint on the same rays: rtk_packet_any2 (rtk_packet_beam2.S -DRTK_ANY), flags written by the kernel"}
            del d_flags
        except Exception as e:      # (kept out of the headline's way)
            any_image = {"error": repr(e)}
        del ref_out
    other_modes = {}
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # the same K steps under the other two treatments of the records, so that one line shows all three: a GPU that
        # traces 16 Grays/s produces ~260 GB/s of records, one xGMI link direction carries ~77 GB/s, and a root takes in at
        # most its 7 links -- what `value` includes decides what bounds it
        for how in ("striped", "root", "none"):
            if how == mode:
                continue
            dist.barrier()
            sync()
            t1 = time.perf_counter()
            for k in range(args.steps):
                step(k, None, how)
            drain()
            sync()
            dist.barrier()
            sync()
            t2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            other_modes[how] = float(t2.item())
    elapsed_no_gather = other_modes.get("none")
    # N > 1 diagnostics: if the scaling is off, WHY -- a slow rank (kernel time per rank: min / max over the ranks of each
    # rank's mean over the timed steps) or an exchange that is not hidden (step time with the exchange minus step time without)
    rank_kernel = None
    if world > 1:
        mine = torch.tensor([float(np.mean(kernel_ms)) if kernel_ms else 0.0], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        rank_kernel = {"per_rank": [round(x, 4) for x in per_rank], "min": round(min(per_rank), 4), "max": round(max(per_rank), 4),
                       "what": "mean over the timed steps of the traversal's time on each rank's launch stream (events around the launch), ms"}
    d_out = d_outs[(args.steps - 1) % len(d_outs)] if args.steps else d_outs[0]

    # ---- the result that was timed ---------------------------------------------------------
    if shadow:
        occ = d_out.cpu().numpy().astype(bool)
        hit_frac = float(occ.mean())
    else:
        rec = d_out.cpu().numpy().view(HIT_RECORD_DTYPE)
        hit_frac = float((rec["prim"] != 0xFFFFFFFF).mean())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # fabric traffic and limiter of the same launch from the committed rocprofv3 PMC summary (separate passes; see
    # scripts/profile_workload.sh + scripts/summarize_profile.py). Dropped when the kernels changed since.
    pj, traffic_src, fresh = (None, None, False) if DRY else load_pmc_summary(args.workload)
    traffic = float(pj["hbm_traffic_bytes_per_launch"]) if (pj and fresh and args.bvh == "device") else None
    limiter = limiter_from_pmc(pj) if (pj and fresh) else {}

    total_rays = n * world * args.steps
    mrays = total_rays / elapsed / 1e6
    k_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    h_frac, h_note, h_fbt, h_real = honest_fractions(achieved, traffic, k_ms, limiter)
    lane_asm = os.environ.get("RTK_AMD_LANE_ASM", "1") != "0" and lane_node_bytes == 64 and not args.static
    beam = os.environ.get("RTK_AMD_PACKET_BEAM", "2")
    kernel_name = ({"0": "rtk_packet_hot", "1": "rtk_packet_beam"}.get(beam, "rtk_packet_beam2") + " (hand-written gfx950 assembly) + rtk_trace_packet_kernel<false> on the tiles it hands back" if packet_kernel else
                   ("rtk_lane_hot_%s (hand-written gfx950 assembly) + " % ("any" if shadow else "closest") if lane_asm else "") +
                   "rtk_trace_kernel<%d, false, false, %s>%s" % (1 if shadow else 0, "true" if lane_node_bytes == 64 else "false", " on the rays it hands back" if lane_asm else ""))
    out = {
        "metric": metric,
        "value": round(mrays, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not DRY else "none: --dry-run-cpu plumbing check, not a measurement",
        "config": {"workload": workload, "rays_per_gpu_per_step": n, "bvh": bvh_kind, "bvh_nodes": info["num_nodes"],
                   "bvh_build_s": round(t_build, 3), "bvh_build_ms_first_call": round(build_ms_first_call, 2),
                   "bvh_build_ms_in_library": round(info["build_ms"], 2),
                   "bvh_build_mtris_s": round(cfg["num_tris"] / max(info["build_ms"], 1e-9) / 1e3, 1) if info["build_ms"] else None,
                   "bvh_build_ms_device_resident_mesh": round(build_ms_device_mesh, 2) if build_ms_device_mesh else None,
                   "bvh_build_mtris_s_device_resident_mesh": round(cfg["num_tris"] / build_ms_device_mesh / 1e3, 1) if build_ms_device_mesh else None,
                   "hit_fraction": round(hit_frac, 4),
                   "gather": {"striped": "every rank receives its stripe of every shard's records over RCCL (batched point-to-point, each link carries 1/N of a shard per step), overlapped with the next step's trace",
                              "root": "all records onto rank 0 over RCCL (batched point-to-point), overlapped with the next step's trace; bound by the root's incoming links",
                              "none": None}[mode],
                   "value_without_gather_mrays_s": round(n * world * args.steps / elapsed_no_gather / 1e6, 2) if elapsed_no_gather else None,
                   "value_root_gather_mrays_s": round(n * world * args.steps / other_modes["root"] / 1e6, 2) if "root" in other_modes else None,
                   "value_striped_gather_mrays_s": round(n * world * args.steps / other_modes["striped"] / 1e6, 2) if "striped" in other_modes else None,
                   "per_rank_kernel_ms": rank_kernel,
                   "exposed_exchange_ms_per_step": ({how: round(((elapsed if how == mode else other_modes[how]) - elapsed_no_gather) / args.steps * 1e3, 4)
                                                     for how in ("striped", "root") if (how == mode or how in other_modes)}
                                                    if (world > 1 and elapsed_no_gather) else None),
                   "without_image_hint": no_hint,
                   "any_hit_on_the_same_image": any_image,
                   "launch": "static" if args.static else "persistent",
                   "ray_order": "RTK_TRACE_SORT_RAYS: re-ordered by origin cell inside every timed step" if args.sort_rays else "as given",
                   "parallelism": "ray-batch shards x%d, BVH replicated" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": h_frac, "traffic": traffic, "frac_by_traffic": h_fbt, "frac_note": h_note, "real_bound": h_real,
                     "traffic_unit": ("bytes per launch of the traversal kernel crossing the L2 -> fabric boundary, Infinity-Cache (MALL) hits "
                                      "INCLUDED: (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 PMC passes, %s; the scene is %.0f MB (the MALL "
                                      "holds 256 MiB), so this is %s" % (traffic_src, info.get("total_device_bytes", 0) / 1e6,
                                      "not pure HBM traffic" if info.get("total_device_bytes", 0) < 256 * 2 ** 20 else "mostly HBM traffic")) if traffic else
                                     ("none: %s was measured on other kernel code (kernel_code_sha16 differs)" % traffic_src if pj else None),
                     "unit_of_work": unit,
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "kernel": kernel_name if not args.sort_rays else kernel_name + " preceded by the re-ordering pre-pass (rtk_ray_entry_keys_kernel, "
                               "3 x k_sort_hist/k_scan_block/k_sort_scatter): kernel_ms is their sum per step",
                     "kernel_ms": round(k_ms, 4),
                     "limiter": dict(limiter, fabric_tb_s=round(traffic / pj["kernel_trace"]["average_ns"] / 1e3, 2) if traffic else None,
                                     note=("VALU pipes busy `valu_busy` of the traversal kernel's time at `valu_lane_utilisation` "
                                           "(SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * time * %.1f GHz), SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)); "
                                           "`salu_issue` = SQ_INSTS_SALU / (256 CUs * time * clock): scalar instructions per cycle of a CU's one scalar unit; "
                                           "`fabric_tb_s` = traffic / kernel time against ~6.3 TB/s achievable HBM and ~8.6 TB/s MALL gather rate; "
                                           "counters from %s" % (CLOCK_GHZ, traffic_src))) if limiter else None,
                     "visits_per_ray": {"nodes": round(ctr["nodes"] / n, 2), "leaves": round(ctr["leaves"] / n, 2),
                                        "triangles": round(ctr["triangles"] / n, 2)},
                     "wave_steps_per_64_rays": {"nodes": round(ctr["wave_node_steps"] * 64.0 / n, 1),
                                                "triangles": round(ctr["wave_triangle_steps"] * 64.0 / n, 1),
                                                "of": "rtk_trace_packet_kernel<true> (C++, one tile per wave, per-lane slab tests)" if packet_kernel else "the counting build of the per-lane kernel"},
                     "timed_kernel_steps": ({"per_pair_of_tiles": {"node_steps": round(pk_ctr["node_steps"] / max(1, pk_ctr["pairs"]), 2),
                                                                  "triangles_fetched": round(pk_ctr["triangles_fetched"] / max(1, pk_ctr["pairs"]), 2),
                                                                  "triangle_group_tests": round(pk_ctr["triangle_group_tests"] / max(1, pk_ctr["pairs"]), 2)},
                                             "pairs": pk_ctr["pairs"], "tiles_handed_back": pk_ctr["tiles_handed_back"],
                                             "handed_back_node_steps": pk_ctr["handed_back_node_steps"], "handed_back_triangle_steps": pk_ctr["handed_back_triangle_steps"],
                                             "counted_by": "rtk_packet_count2 (rtk_packet_beam2.S -DRTK_COUNT)"} if pk_ctr else None),
                     "per_ray_model": {"bytes_per_ray": round(per_ray_bytes / n, 1), "gb_s": round(per_ray_bytes / (k_ms * 1e-3) / 1e9, 1),
                                       "note": "every lane's visit priced at full size; caches absorb these for coherent rays, so this is "
                                               "cache bandwidth demand, not a fraction of the HBM roofline"},
                     "kernel_mrays_s": round(n / (k_ms * 1e-3) / 1e6, 1)},
    }
    if args.bvh == "device" and build_ms_device_mesh:
        # SURVEY.md section 8d, build formula with this build's sizes: 36 B positions in, the sort item out (< 2^24 triangles:
        # one 8-B word = 40-bit code over the index, 5 passes; else a 12-B (key, index) pair, 8 passes), P passes x item read +
        # written, 2 x 32 B binary node (refit write, collapse read), 128 B x wide nodes per triangle + 48 B triangle record out
        # (key width from n: ceil(log2 n) + 8 bits in whole 8-bit passes, 3 ... 5 of them for the 8-byte words; rtk_build.hip)
        lg = int(np.ceil(np.log2(max(2, cfg["num_tris"]))))
        passes, item = (min(5, max(3, (lg + 8 + 7) // 8)), 8) if cfg["num_tris"] < (1 << 24) else (8, 12)
        bpt = 36 + item + passes * 2 * item + 64 + 128.0 * info["num_nodes"] / cfg["num_tris"] + 48
        gbs = cfg["num_tris"] * bpt / (build_ms_device_mesh * 1e-3) / 1e9
        out["build"] = {"triangles": cfg["num_tris"], "ms": round(build_ms_device_mesh, 3), "what": "rtk_dev_scene_build, mesh resident in HBM, "
                        "wall time inside the library (all kernels, no PCIe)", "sort_passes": passes,
                        "algorithmic_bytes_per_triangle": round(bpt, 1), "achieved_gb_s": round(gbs, 1),
                        "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "mtris_s": round(cfg["num_tris"] / build_ms_device_mesh / 1e3, 1)}

    # ---- CPU baseline: the oracle (a port of rtk.c's trace path) on this host's cores -----
    if not args.no_cpu_baseline:
        # idle OpenMP workers sleep instead of spinning: the pool grows to every core for the all-cores figure, and on a box
        # whose cgroup grants fewer CPUs than it shows, spinning workers would eat the quota of the timed ones
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        from oracle import pyoracle
        t0 = time.time()
        if oracle_blob is None:
            oracle_blob = pyoracle.build_scene([dict(positions=tris)])
        t_cpu_build = time.time() - t0
        threads = args.cpu_threads if args.cpu_threads > 0 else pyoracle.default_threads()
        try:
            all_cores = len(os.sched_getaffinity(0))
        except AttributeError:
            all_cores = os.cpu_count() or 1
        all_cores = max(1, min(all_cores, pyoracle.lib().ora_max_threads()))
        # what the container may actually USE (cgroup v2 cpu.max = "quota period"): a one-GPU box shows 128+ CPUs and grants ~16
        cpu_quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            cpu_quota = None if q == "max" else round(float(q) / float(per), 2)
        except Exception:
            try:        # cgroup v1
                q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                cpu_quota = round(q / per, 2) if q > 0 else None
            except Exception:
                pass
        # bounded sample: every k-th ray of the same batch (a prefix would be all top-of-frame misses).
        # The timed calls write 16-byte records (what the GPU writes) into a buffer that was allocated and touched beforehand;
        # blob, rays and output live in memory whose pages the worker threads first-touched round robin (a multi-socket host:
        # not all on one memory controller). Round 3's figures (46.6 Mrays/s on 16 threads, 33.4 on 128) timed the first touch
        # of 1.1 GB of 68-byte rtk_hit output inside the call, handed out in dynamic chunks of 64 rays.
        blob_s = pyoracle.SpreadBuffer(oracle_blob.size, oracle_blob, threads=all_cores)
        probe_sel = np.arange(0, n, max(1, n >> 17))
        probe_rays = np.ascontiguousarray(rays[probe_sel])
        probe_out = pyoracle.SpreadBuffer(len(probe_sel) * 16, threads=all_cores)
        pyoracle.trace_records(blob_s, probe_rays[:4096], probe_out, threads=all_cores)      # spins the thread pool up
        t0 = time.time()
        pyoracle.trace_records(blob_s, probe_rays, probe_out, threads=threads)
        rate = len(probe_sel) / max(time.time() - t0, 1e-6)
        stride = 1
        while n // stride > rate * args.cpu_seconds and stride < n:
            stride *= 2
        sel = np.arange(0, n, stride)
        sample = len(sel)
        rays_s = pyoracle.SpreadBuffer(sample * 32, rays if stride == 1 else np.ascontiguousarray(rays[sel]), threads=all_cores)
        sample_rays = rays_s.view(rays.dtype)[:sample]
        out_s = pyoracle.SpreadBuffer(sample * 16, threads=all_cores)
        t0 = time.time()
        rec_cpu = pyoracle.trace_records(blob_s, rays_s, out_s, threads=threads).copy()
        dt = time.time() - t0
        ohits, omask = rec_cpu, rec_cpu["triangle_index"] != 0xFFFFFFFF
        t0 = time.time()
        pyoracle.trace_records(blob_s, probe_rays, probe_out, threads=1)
        rate1 = len(probe_sel) / max(time.time() - t0, 1e-6)
        # the same sample on every core this process may use ("all host cores, count stated")
        rate_all = None
        if all_cores != threads:
            t0 = time.time()
            pyoracle.trace_records(blob_s, rays_s, out_s, threads=all_cores)
            rate_all = sample / max(time.time() - t0, 1e-6)

        def parity(oh, om):
            if shadow:
                return {"rays": sample, "occluded_flag_mismatches": int((occ[sel] != om).sum())}
            g = rec[sel]
            gm = g["prim"] != 0xFFFFFFFF
            both = gm & om
            mism = int((g["prim"][both] != oh["triangle_index"][both]).sum()) + int((gm != om).sum())
            same = both & (g["prim"] == oh["triangle_index"])   # t/u/v are compared where both picked the same triangle
            # a relative tolerance means nothing where t itself is what a cancellation leaves (a ray origin on a triangle): hits at
            # |t| < 1e-6 are counted apart; they are pinned EXACTLY instead (the real rtk.c's values under both leaf groupings,
            # tests/golden/tiny_t.npz, test_tiny_t_rays_are_reference_answers)
            tiny = same & (np.abs(oh["t"]) < 1e-6)
            same_n = same & ~tiny
            relv = np.abs(g["t"][same_n] - oh["t"][same_n]) / np.abs(oh["t"][same_n]) if same_n.any() else np.zeros(1)
            rel = float(relv.max())
            t_at = float(oh["t"][same_n][int(relv.argmax())]) if same_n.any() else 0.0
            tiny_rel = float((np.abs(g["t"][tiny] - oh["t"][tiny]) / np.abs(oh["t"][tiny])).max()) if tiny.any() else 0.0
            exact = float(np.mean((g["t"][same] == oh["t"][same]) & (g["u"][same] == oh["u"][same]) &
                                  (g["v"][same] == oh["v"][same]))) if same.any() else 1.0
            # every ray on which the two disagree about WHICH triangle, with both candidates' t: these are near ties
            # (two triangles within an ulp in t) that rtk.c's group-of-four double-precision rule (rtk.c:302-336)
            # resolves by leaf grouping; tests/golden/near_ties.npz holds the reference's values for both groupings
            bad = np.nonzero((gm != om) | (both & (g["prim"] != oh["triangle_index"])))[0]
            listing = [{"ray": int(sel[i]), "gpu_prim": int(g["prim"][i]), "gpu_t": float(g["t"][i]),
                        "cpu_prim": int(oh["triangle_index"][i]) if om[i] else None, "cpu_t": float(oh["t"][i]) if om[i] else None,
                        "t_ulps_apart": f32_ulps(g["t"][i], oh["t"][i]) if (gm[i] and om[i]) else None} for i in bad[:64]]
            return {"rays": sample, "ids_exact": mism == 0, "id_mismatches": mism, "mismatching_rays": listing, "max_rel_t": rel,
                    "t_at_max_rel_t": t_at, "tuv_bit_exact_fraction": exact,
                    "hits_below_1e-6": {"rays": int(tiny.sum()), "max_rel_t": tiny_rel,
                                        "note": "pinned bit for bit by tests/golden/tiny_t.npz (real rtk.c, both leaf groupings), not by a relative tolerance"}}

        base = {"value": round(sample / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": "%d rays = every %d-th ray of the same batch, closest-hit on the oracle's SAH BVH4 (built in %.1fs), %d OpenMP threads, 16-byte records into "
                          "pre-touched memory, blob / rays / output pages first-touched round robin by the workers; 1 thread: %.3f Mrays/s"
                          % (sample, stride, t_cpu_build, threads, rate1 / 1e6),
                "one_thread": {"value": round(rate1 / 1e6, 3), "unit": "Mrays/s", "cores": 1},
                "host_cpus": os.cpu_count(), "cgroup_cpu_quota_cores": cpu_quota,
                "scaling_note": ("the container's CPU quota is %.1f cores: more threads than that share the same CPU time, so the all-cores figure cannot "
                                 "exceed ~%.1f x the one-thread figure" % (cpu_quota, cpu_quota)) if cpu_quota else
                                "no cgroup CPU quota: the all-cores figure is bound by the host (memory bandwidth of the BVH gathers, SMT siblings)",
                "all_cores": {"value": round(rate_all / 1e6, 3), "cores": all_cores, "unit": "Mrays/s",
                              "what": "the same sample on every core this process may use (os.sched_getaffinity)"} if rate_all else
                             {"value": round(sample / dt / 1e6, 3), "cores": threads, "unit": "Mrays/s", "what": "the main figure already uses every core"}}
        # (1) the CPU's own BVH: another valid BVH of the same triangles (near-ties may resolve differently)
        base["parity_vs_gpu_oracle_bvh"] = parity(ohits, omask)
        # (2) the SAME BVH: the oracle traverses the blob exported from the GPU-built scene -> bit-exact
        if args.bvh == "device":
            exported = pyoracle.Blob(ds.export_blob())
            eh = pyoracle.trace_records(exported, rays_s, out_s, threads=all_cores).copy()
            base["parity_vs_gpu_same_bvh"] = parity(eh, eh["triangle_index"] != 0xFFFFFFFF)
        else:
            base["parity_vs_gpu_same_bvh"] = base["parity_vs_gpu_oracle_bvh"]
        for b_ in (blob_s, probe_out, rays_s, out_s):
            b_.free()
        out["cpu_baseline"] = base
    if world == 1 and args.workload == "coherent" and args.bvh == "device" and not args.no_other_workloads and not DRY and W == 4096:
        # BASELINE.json configs 3 and 5 in the same run, so that their numbers are observed by whoever runs the default
        # command: compact lines, a few steps each, no CPU leg
        ds.free()
        del d_rays, d_outs
        torch.cuda.empty_cache()
        out["other_workloads"] = {}
        for kind in ("incoherent", "shadow"):
            try:
                out["other_workloads"][kind] = other_workload(kind, steps=max(3, min(args.steps, 20)), warmup=5)
            except Exception as e:      # the headline line must not die with an extra
                out["other_workloads"][kind] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
