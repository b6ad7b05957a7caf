#!/usr/bin/env python3
"""bench.py -- images/sec through CSNet-100K (csnet-L-x2) eval forward on synthetic 3x224x224 batches.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script re-executes itself under ``python -m torch.distributed.run`` with N
ranks on 127.0.0.1 (one process per GPU over RCCL); launched by torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.

A step = one pass of the hot path (``CSNet.forward``, csnet.py:365-387) over one batch of 64 images per
GPU that is already resident in HBM (BASELINE.json configs[1]: fp32 forward, batch 64, 1 MI355X).  The path
shards by image with no data-path collective, so N>1 runs N independent shards ("weak" scaling); the only
cross-rank operations are the barrier around the timed region and the MAX-reduce of the elapsed time.

Rank 0 prints ONE JSON line with, besides the driver's contract fields:
  roofline      for the kernel that dominates the step: algorithmic bytes per launch (unit inputs read
                once + outputs written once) / mean launch duration measured with HIP
                events on the launch stream, against the 8 TB/s HBM3E peak;
  cpu_baseline  the CPU oracle (a port of the reference path onto the same ATen CPU kernels) timed on the
                host cores of this box on a bounded sample (rank 0, N=1 only): batch-8 throughput on all threads
                (``value``), the test.py-style batch-1 latency loop, the 1-thread figure and the CPU model;
  self_check    max |y - oracle| of the first two images of the batch the timed loop just produced (the timed path is
                the hipGraph replay; the run FAILS if it exceeds 1e-4).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_IMAGE = 169_654_464   # csnet-L-x2 fp32 224x224: sum of csn_unit_algorithmic_bytes (SURVEY.md 8(d) says 169,704,640:
                                    # it counts the 50,176 B final map twice); only used when the profile is skipped


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(man, batch=8, target_s=20.0):
    """Time the CPU oracle on a bounded sample (about 20-30 s of CPU work in total): batch-8 throughput and
    the reference caller's batch-1 loop (test.py:87-93) at several thread counts.  ``value`` is the BEST throughput found
    (on a 128-core box the oneDNN kernels of this small network are fastest far below all threads), ``cores`` the threads
    it used; every measured point is listed."""
    from oracle import csnet_oracle as O, inputs as I
    sd = O.load_weights(man)
    lc = O.load_layer_config_json(man)
    x = torch.from_numpy(I.randn_batch(0, batch))
    all_threads = torch.get_num_threads()
    counts = sorted({1, min(8, all_threads), min(16, all_threads), min(32, all_threads), all_threads})

    def loop(xx, budget, nmax):
        O.csnet_forward(lc, sd, xx)          # warm-up
        ts = []
        t_end = time.perf_counter() + budget
        while len(ts) < nmax and (len(ts) < 2 or time.perf_counter() < t_end):
            t0 = time.perf_counter()
            O.csnet_forward(lc, sd, xx)
            ts.append(time.perf_counter() - t0)
        return ts

    points = []
    t_begin = time.perf_counter()
    with torch.no_grad():
        for nthr in counts:
            torch.set_num_threads(nthr)
            t8 = loop(x, target_s / (2 * len(counts)), 10)
            t1 = loop(x[:1], target_s / (2 * len(counts)), 10)
            points.append({"threads": nthr, "batch8_images_per_sec": round(batch / statistics.median(t8), 3),
                           "batch1_latency_ms": round(statistics.median(t1) * 1e3, 2), "forwards": len(t8) + len(t1)})
        torch.set_num_threads(all_threads)
    best = max(points, key=lambda p: max(p["batch8_images_per_sec"], 1e3 / p["batch1_latency_ms"]))
    best_v = max(best["batch8_images_per_sec"], 1e3 / best["batch1_latency_ms"])
    return dict(value=round(best_v, 3), unit="images/sec", cores=best["threads"], kind="port",
                cpu_model=_cpu_model(), host_threads=all_threads, points=points,
                sample=f"{sum(p['forwards'] for p in points)} eval forwards (batch {batch} and batch 1, 3x224x224, seed 0) "
                       f"through oracle/csnet_oracle.py (torch {torch.__version__} CPU ops) at {counts} threads, "
                       f"{time.perf_counter() - t_begin:.1f} s; value = best point")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def respawn(n):
    """``python bench.py --gpus N`` from a bare shell: run N ranks of this script under torch.distributed.run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this driver)
    env["SOD100K_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def train_algorithmic_bytes(model):
    """(3 * in + 7 * out) activation elements per unit (two-pass BN forward, backward reads dy and z twice,
    reads x and writes dx) x 4 bytes, per image at 224 x 224."""
    arena = model._ensure_arena()
    units, acts, _ = model.describe(arena.offsets)
    tin = tout = 0
    for u in units:
        for i in range(int(u.n_in)):
            if u.cin[i] > 0 and u.in_act[i] >= 0:
                c, lvl = acts[u.in_act[i]]
                tin += c * (224 >> lvl) * (224 >> lvl)
        if u.kind == 4:                      # cls_layer: its output is the full-resolution logit map
            tout += 224 * 224
            continue
        for j in range(int(u.n_out)):
            if u.cout[j] > 0 and u.out_act[j] >= 0:
                c, lvl = acts[u.out_act[j]]
                tout += c * (224 >> lvl) * (224 >> lvl)
    return 4 * (3 * tin + 7 * tout)


def measured_copy_peak(dev, gib=1.0, iters=10):
    """Device-to-device copy bandwidth on this board, GB/s of (read + written) bytes: the achievable HBM figure next to the spec.
    The library's own streaming kernel (csn_stream_copy: 128-bit accesses, four in flight per lane) and torch's copy_; the best."""
    try:
        from sod100k_amd import _native as N
        lib = N.load()
        n = int(gib * 2 ** 30) // 4
        a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        stream = torch.cuda.current_stream(dev).cuda_stream
        best = 0.0
        for fn in (lambda: N.check(lib, lib.csn_stream_copy(a.data_ptr(), b.data_ptr(), n, stream), "csn_stream_copy"),
                   lambda: b.copy_(a)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            best = max(best, 2 * n * 4 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
        return round(best, 1)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--sub-batch", type=int, default=int(os.environ.get("CSN_SUB_BATCH", "0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=5)
    ap.add_argument("--no-fuse-cls", action="store_true",
                    help="keep cls_layer as its own launch (PMC calibration: it reads exactly 79x112x112x4 B per image)")
    ap.add_argument("--train-batch", type=int, default=256,
                    help="images per GPU per train step (SURVEY 8(d) config 3: 256; 0 = --batch)")
    ap.add_argument("--csf-batch", type=int, default=32,
                    help="also time the CSF+Res2Net forward (BASELINE config 5: batch x 3x352x352) on 1 GPU; 0 = skip")
    ap.add_argument("--csf-steps", type=int, default=5)
    ap.add_argument("--train-steps", type=int, default=10,
                    help="also time this many full train steps (0 = skip); reported under \"train_step\"")
    ap.add_argument("--no-train-bf16", action="store_true", help="skip the bf16-activation train step (\"train_step_bf16\")")
    ap.add_argument("--train-net", default="x2+unpruned", choices=["x2", "x2+unpruned", "unpruned"],
                    help="networks of the train-step points: the shipped csnet-L-x2 (BASELINE config 3) and / or the UN-PRUNED expand 2.0, "
                         "basic_split [0.5, 0.5] net the reference's training recipe starts from (csnet-L-x2_train.yml:9-18), batch 64")
    ap.add_argument("--unpruned-batch", type=int, default=256,
                    help="images per GPU of the un-pruned net's bf16 point (BASELINE config 3's batch); its fp32 point stays at 64")
    ap.add_argument("--no-latency-b1", action="store_true", help="skip the batch-1 latency loop (\"latency_b1\"): kernel traces of the "
                    "headline workload then hold batch-64 launches only")
    ap.add_argument("--event-steps", type=int, default=50,
                    help="extra steps timed one by one with HIP events after the contract's timed region (median reported)")
    ap.add_argument("--emu-plumbing", action="store_true",
                    help="CPU-only test of the launch / rendezvous / reporting plumbing (tests/test_dist_cpu.py): kernels "
                         "emulated by tests/emu on gloo, tiny shapes; the line is marked invalid and is never a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    emu = args.emu_plumbing
    if emu:
        dev = torch.device("cpu")
        args.batch, args.train_batch, args.csf_batch, args.profile_iters = 2, 2, 0, 1
        args.no_cpu_baseline = True
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: no ROCm device {local_rank} (visible: {torch.cuda.device_count()})")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    from sod100k_amd import dist as D
    world = D.init(device=dev)
    nranks = world
    if D.active():
        import torch.distributed as tdist
        nranks = tdist.get_world_size()
        probe = torch.ones(1, device=dev)
        tdist.all_reduce(probe)               # the communicator (RCCL on GPUs) really spans `nranks` processes
        assert int(probe.item()) == nranks, (float(probe.item()), nranks)
    S = 32 if emu else 224

    from sod100k_amd.model import csnet as M
    from sod100k_amd.checkpoint import load_manifest_state_dict
    from sod100k_amd import _native as _N
    N_SRC_SHA = _N.sources_sha16()
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    model = M.build_model(predefine=man)
    sd = load_manifest_state_dict(man)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    model._sub_batch = args.sub_batch
    if emu:
        import ctypes
        from sod100k_amd import _native as N_
        model._lib = N_.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libcsnet_emu.so")))
    sync = (lambda: None) if emu else (lambda: torch.cuda.synchronize(dev))

    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(rank)
    x_host = torch.randn(B, 3, S, S, generator=g)
    x = x_host.to(dev)                                          # synthetic, resident in HBM before timing
    eng = model.engine_for(x)
    eng.refresh(model._arena.flat)
    if args.no_fuse_cls:
        from sod100k_amd import _native as N
        eng.set_option(N.OPT_FUSE_CLS, 0)
    y = torch.empty(B, 1, S, S, device=dev)

    def step():
        eng.forward(x, out=y)

    for _ in range(args.warmup):
        step()
    dt = D.timed_region(step, args.steps, sync=sync, device=dev)

    # ---- the bench checks its own output: the timed path is the hipGraph replay, y is what it just wrote ----
    self_check = None
    if rank == 0:
        from oracle import csnet_oracle as O
        with torch.no_grad():
            ref = O.csnet_forward(O.load_layer_config_json(man), O.load_weights(man), x_host[:2])
        err = float((y[:2].cpu() - ref).abs().max())
        self_check = {"max_abs_vs_oracle": err, "images": 2, "tol": 1e-4, "path": "output of the last timed step"}
        if not (err <= 1e-4) and os.environ.get("SOD100K_BENCH_KNOCKOUT") != "1":   # knock-out builds compute garbage on purpose
            raise SystemExit(f"bench self-check FAILED: max|y - oracle| = {err:.3e} > 1e-4")

    # ---- per-step HIP events on the launch stream (median of >= 50 single steps) ----
    ev_stats = None
    if not emu and args.event_steps > 0:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.event_steps)]
        for a_, b_ in evs:                     # eng.forward launches on torch's current stream (engine._stream)
            a_.record(); step(); b_.record()
        torch.cuda.synchronize(dev)
        ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        ev_stats = {"steps": len(ts), "median_ms": round(statistics.median(ts), 4), "min_ms": round(ts[0], 4),
                    "p90_ms": round(ts[int(0.9 * (len(ts) - 1))], 4),
                    "images_per_sec_at_median": round(B / (statistics.median(ts) * 1e-3), 1)}

    roofline, total_alg = None, ALG_BYTES_PER_IMAGE * B
    eng_p = None
    if not emu:
        # ---- per-kernel roofline: a HIP event after EVERY kernel launch, on the launch stream.  The timed region runs the product's
        # default plan: at batch >= 32 that is two half-batch slices side by side on two stream lanes (round 6), where launches
        # overlap and a launch interval is not a kernel duration.  The per-kernel figures are therefore taken on the SAME kernels
        # launched once per whole batch on one stream (CSN_SLICE_LANES=0 plan: the launches profiles/*_kernel_stats_eval_nolanes.md
        # traces); the slices' own serialised per-launch figures are reported next to them (`timed_region_launches`) ----
        eng_p = eng
        sliced = None
        if getattr(eng, "slice_lanes", False):
            ms_s, names_s, nbytes_s = eng.profile(x, iters=args.profile_iters)
            ks = eng.kernel_stats()
            eng_p = model.engine_for(x, slice_lanes=False)
            eng_p.refresh(model._arena.flat)
            if args.no_fuse_cls:
                eng_p.set_option(_N.OPT_FUSE_CLS, 0)
            yp = torch.empty_like(y)
            for _ in range(3):
                eng_p.forward(x, out=yp)
            wb = D.timed_region(lambda: eng_p.forward(x, out=yp), args.steps, sync=sync, device=dev)
            sliced = dict(slices=(B + eng.sub_batch - 1) // eng.sub_batch, images_per_slice=eng.sub_batch,
                          serialised_events_ms=round(sum(ms_s), 3),
                          per_kernel={k: dict(ms=round(v[0], 3), launches=v[1], us_per_launch=round(v[0] * 1e3 / v[1], 2))
                                      for k, v in ks.items()},
                          whole_batch_one_stream=dict(ms_per_step=round(wb / args.steps * 1e3, 4),
                                                      images_per_sec=round(B * args.steps / wb, 1)))
            del yp
        ms, names, nbytes = eng_p.profile(x, iters=args.profile_iters)
        kstats = eng_p.kernel_stats()                      # kernel name -> (ms per forward, launches per forward)
        agg = {}
        for n, nb in zip(names, nbytes):                   # algorithmic bytes of the units each kernel implements
            agg.setdefault(n, dict(bytes=0))["bytes"] += nb
        for n, (kms, kl) in kstats.items():
            agg.setdefault(n, dict(bytes=0)).update(ms=kms, launches=kl)
        dom = max((k for k in agg if agg[k]["bytes"] > 0 and "ms" in agg[k]), key=lambda k: agg[k]["ms"])
        d = agg[dom]
        bytes_per_launch = d["bytes"] / d["launches"]
        us_per_launch = d["ms"] * 1e3 / d["launches"]
        bracket_us = eng_p.profile_bracket_us()
        achieved = bytes_per_launch / (us_per_launch * 1e-6) / 1e9
        # HBM traffic of the dominant kernel: NOT a quantity of this run -- rocprofv3 cannot run inside the bench; it is the
        # per-launch average of the committed counter passes (tools/gpu_pmc_hbm.sh, FETCH_SIZE / WRITE_SIZE calibrated with
        # tools/probes/fetch_cal), labelled with where it came from
        traffic = traffic_src = None
        traffic_tree_ok = False
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                fwd_bytes = pj.get(dom, {}).get("hbm_bytes_per_forward")
                if fwd_bytes:
                    traffic = int(fwd_bytes / d["launches"])       # per launch of THIS run's launch count
                traffic_src = "profiles/pmc_latest.json: " + pj.get("_source", "")
                traffic_tree_ok = pj.get("_kernel_sources_sha16") == N_SRC_SHA
            except Exception:
                traffic = None
        total_alg = sum(nbytes)
        total_ms = sum(v["ms"] for v in agg.values() if "ms" in v)
        # bytes a fused launch group really moves where that is less than its units' algorithmic sum: a whole ILBlock
        # on ilb_kernel = three consecutive units, block input + block output only: (in1 + out1) + (in3 + out3) - (in2 + out2)
        moved = {}
        ilb_units = [i for i, n in enumerate(names) if n == "ilb_kernel"]
        for i in ilb_units[::3]:
            if i + 2 < len(nbytes) and names[i + 1] == names[i + 2] == "ilb_kernel":
                moved["ilb_kernel"] = moved.get("ilb_kernel", 0) + nbytes[i] + nbytes[i + 2] - nbytes[i + 1]
        # counter bytes of the WHOLE forward (every kernel family of the committed counter pass) next to the algorithmic figure
        counter_total = None
        try:
            if os.path.exists(pmc):
                pj_all = json.load(open(pmc))
                counter_total = int(sum(v.get("hbm_bytes_per_forward", 0) for k, v in pj_all.items() if isinstance(v, dict) and not k.startswith("_")))
        except Exception:
            counter_total = None
        peak_meas = measured_copy_peak(dev)
        roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4),
                        # the on-box figure next to the 8 TB/s spec -- a device-to-device copy of 1 GiB in this process
                        # (read + written bytes / time); what a pure streaming kernel reaches on this board
                        peak_measured=peak_meas, frac_of_measured=(round(achieved / peak_meas, 4) if peak_meas else None),
                        traffic=traffic, traffic_source=traffic_src,
                        # the counter file was collected on exactly these kernel sources (sha256 over csrc/): else it is a stale number
                        traffic_from_this_tree=bool(traffic_tree_ok), kernel_sources_sha16=N_SRC_SHA,
                        bytes_per_launch=int(bytes_per_launch), us_per_launch=round(us_per_launch, 2),
                        launches_per_step=d["launches"],
                        event_bracket_us=round(bracket_us, 2),   # subtracted from every per-launch interval (measured on empty launches)
                        # ... and the same figures WITHOUT that correction: the raw event-to-event interval per launch
                        us_per_launch_raw=round(us_per_launch + bracket_us, 2),
                        frac_raw=round(bytes_per_launch / ((us_per_launch + bracket_us) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        launches_profiled=("whole batch on one stream (CSN_SLICE_LANES=0 plan)" if sliced else "the timed region's own"),
                        timed_region_launches=sliced,
                        # the whole forward as TIMED (value): algorithmic bytes / ms_per_step -- with slices on two lanes this is
                        # above the serialised `whole_step` figure below (overlap), and it is what the img/s number stands on
                        timed_step=dict(achieved=round(total_alg / (dt / args.steps) / 1e9, 1),
                                        frac=round(total_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)),
                        whole_step=dict(algorithmic_bytes=int(total_alg), events_ms=round(total_ms, 3),
                                        achieved=round(total_alg / (total_ms * 1e-3) / 1e9, 1),
                                        frac=round(total_alg / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        # the bytes the counters saw for one forward (profiles/pmc_latest.json, all families) / the same time
                                        counter_bytes=counter_total,
                                        counter_frac=(round(counter_total / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if counter_total else None),
                                        counter_from_this_tree=bool(traffic_tree_ok)),
                        per_kernel={k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                                            us_per_launch=round(v["ms"] * 1e3 / v["launches"], 2),
                                            alg_GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None,
                                            # the fused depthwise pair implements TWO units per launch: its algorithmic
                                            # bytes count both units' (in + out), the kernel physically moves half of that
                                            **({"moved_GBps": round(0.5 * v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                                               if k in ("dw3x3x2_bn_prelu_kernel", "dw3x3x2_fast_kernel") and v["bytes"] else {}),
                                            **({"moved_GBps": round(moved[k] / (v["ms"] * 1e-3) / 1e9, 1)} if k in moved else {}))
                                    for k, v in agg.items() if "ms" in v})

    sub_b = eng_sub(eng, B)
    lat_b1 = None
    if not emu and rank == 0 and not args.no_latency_b1:
        try:
            lat_b1 = latency_b1(model, dev)
        except Exception as e:       # a secondary data point must never take the headline line down
            lat_b1 = {"error": f"{type(e).__name__}: {e}"}
    # ---- second data point: the full train step (fwd train-mode + BCE + backward + gradient all-reduce + Adam) ----
    train = train16 = unpruned = None
    if args.train_steps > 0:
        from sod100k_amd.tools.train import FusedTrainer
        del eng, y
        eng_p = None
        model._engines = {}
        torch.cuda.empty_cache()
        model.train()
        model.flops_hook(1.0)                       # csnet-L-x2_train.yml: FLOPS.EXPAND 1.0, WEIGHT 3.0
        TB = args.train_batch or B
        model.set_batchsize(TB)
        xt = x if TB == B else torch.randn(TB, 3, S, S, generator=g).to(dev)
        tgt = (torch.rand(TB, 1, S, S, generator=g) > 0.5).float().to(dev)

        def time_train(act_dtype):
            """act_dtype "fp32": the reference's arithmetic; "bf16": BASELINE config 3 (bfloat16 activation storage)."""
            esz = 2 if act_dtype == "bf16" else 4
            tr = FusedTrainer(model, lr=1e-4, weight_decay=5e-3, flops_weight=3.0, batchsize=TB, act_dtype=act_dtype)

            def tstep():
                tr.step(xt, tgt, world_size=world)

            for _ in range(3):
                tstep()
            tdt = D.timed_region(tstep, args.train_steps, sync=sync, device=dev)
            t_alg = train_algorithmic_bytes(model) // 4 * esz * TB if not emu else 0
            t_bw = t_alg / (tdt / args.train_steps) / 1e9
            # HBM traffic of one step: the committed counter passes (tools/gpu_pmc_train.sh), not a quantity of this run
            t_traffic = t_src = None
            t_tree_ok = False
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_train_latest.json")))
                if TB == 256 and act_dtype in pj:
                    t_traffic = int(pj[act_dtype]["hbm_bytes_per_step"])
                    t_src = "profiles/pmc_train_latest.json: " + pj.get("_source", "")
                    t_tree_ok = pj.get("_kernel_sources_sha16") == N_SRC_SHA
            except Exception:
                pass
            rec = {"value": round(world * TB * args.train_steps / tdt, 1), "unit": "images/sec",
                   "ms_per_step": round(tdt / args.train_steps * 1e3, 3), "steps": args.train_steps,
                   "batch_per_gpu": TB, "dtype": "f32" if esz == 4 else "bf16 activations, f32 arithmetic / parameters / optimizer",
                   "loss": round(float(tr.loss), 6),
                   "roofline": {"bound": "hbm", "achieved": round(t_bw, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(t_bw / HBM_PEAK_GBS, 4), "bytes_per_step": int(t_alg),
                                "traffic": t_traffic, "traffic_source": t_src, "traffic_from_this_tree": bool(t_tree_ok),
                                "what": f"algorithmic (3*in + 7*out) x {esz} B per unit (SURVEY 8(d)) / whole-step time"},
                   "what": "train-mode forward (batch-stat BN + penalty) + BCE + backward + "
                           + ("RCCL all-reduce of the flat gradient + " if world > 1 else "") + "Adam, csnet-L-x2 weights"}
            del tr
            model._engines = {}
            torch.cuda.empty_cache()
            return rec

        if "x2" in args.train_net.split("+"):
            train = time_train("fp32")
            if not args.no_train_bf16:
                train16 = time_train("bf16")
        model.set_train_act_dtype("fp32")
        model.eval()
        # ---- secondary point (VERDICT r3 #4a): the network the reference actually TRAINS -- un-pruned, expand 2.0,
        # basic_split [0.5, 0.5], 788,631 parameters, random init -- one full step at batch 64 per GPU in both storage modes
        if "unpruned" in args.train_net.split("+") and not emu:
            try:
                import contextlib, io
                with contextlib.redirect_stdout(io.StringIO()):
                    um = M.build_model(basic_split=[0.5, 0.5], expand=2.0, save_path="/tmp")
                um = um.to(dev).train()
                um.flops_hook(1.0)
                unpruned = {"what": "un-pruned training network (expand 2.0, basic_split [0.5, 0.5], csnet-L-x2_train.yml:9-18), random init: "
                                    "train-mode forward + BCE + backward + Adam; bf16 storage at config 3's batch, fp32 at 64",
                            "parameters": int(sum(p.numel() for p in um.parameters()))}
                u_alg = train_algorithmic_bytes(um) // 4       # (3 in + 7 out) activation ELEMENTS per image, SURVEY 8(d)
                try:
                    u_traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_train_unpruned_latest.json")))
                except Exception:
                    u_traffic = {}
                for adt, UB in (("fp32", min(64, args.unpruned_batch)), ("bf16", args.unpruned_batch)):
                    esz = 2 if adt == "bf16" else 4
                    um.set_batchsize(UB)
                    ux = torch.randn(UB, 3, S, S, generator=g).to(dev)
                    ut = (torch.rand(UB, 1, S, S, generator=g) > 0.5).float().to(dev)
                    utr = FusedTrainer(um, lr=1e-4, weight_decay=5e-3, flops_weight=3.0, batchsize=UB, act_dtype=adt)
                    for _ in range(3):
                        utr.step(ux, ut, world_size=world)
                    n_ = max(3, args.train_steps // 2)
                    udt = D.timed_region(lambda: utr.step(ux, ut, world_size=world), n_, sync=sync, device=dev)
                    u_bw = u_alg * esz * UB / (udt / n_) / 1e9
                    unpruned[adt] = {"ms_per_step": round(udt / n_ * 1e3, 3), "images_per_sec": round(world * UB * n_ / udt, 1),
                                     "batch_per_gpu": UB,
                                     "loss": (round(float(utr.loss), 6) if np.isfinite(float(utr.loss)) else None),
                                     "roofline": {"bound": "hbm", "achieved": round(u_bw, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": round(u_bw / HBM_PEAK_GBS, 4), "bytes_per_step": int(u_alg * esz * UB),
                                                  "traffic": u_traffic.get(adt, {}).get("hbm_bytes_per_step") if UB == (64 if adt == "fp32" else 256) else None,
                                                  "traffic_source": u_traffic.get("_source"),
                                                  "traffic_from_this_tree": u_traffic.get("_kernel_sources_sha16") == N_SRC_SHA,
                                                  "dominant_kernels": ({k: round((v["read_bytes"] + v["write_bytes"]) / 1e9, 2)
                                                                        for k, v in list(u_traffic.get(adt, {}).get("by_kernel", {}).items())[:4]}
                                                                       or None),
                                                  "what": f"algorithmic (3*in + 7*out) x {esz} B per unit of THIS net (SURVEY 8(d)) / whole-step time; "
                                                          "traffic / dominant_kernels (GB per step): the committed counter pass"}}
                    del utr, ux, ut
                    um._engines = {}
                    torch.cuda.empty_cache()
                del um
            except Exception as e:          # a secondary data point must never take the headline line down
                unpruned = {"error": f"{type(e).__name__}: {e}"}

    csf = None
    if world == 1 and args.csf_batch > 0:
        try:
            csf = csf_point(dev, args.csf_batch, args.csf_steps)
        except Exception as e:          # a secondary data point must never take the headline line down
            csf = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "images/sec CSNet-100K 3x224x224 fwd (and fwd+bwd) at 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not emu else "INVALID: CPU plumbing run with emulated kernels (tests only)",
            "nranks": nranks, "backend": ("gloo" if emu else "nccl (RCCL)") if D.active() else "none",
            "rccl_env": D.rccl_env(),
            "config": {"workload": "CSNet-100K (csnet-L-x2 shipped checkpoint) fp32 eval forward, "
                                   f"batch {B} x 3x{S}x{S} per GPU, inputs resident in HBM",
                       "batch_per_gpu": B, "global_batch": B * world, "sub_batch": sub_b,
                       "launch_schedule": (f"{(B + sub_b - 1) // sub_b} batch slices of {sub_b} images side by side on the plan's stream "
                                           "lanes, one hipGraph replay per step" if sub_b < B else "whole batch per launch, one hipGraph replay per step"),
                       "parallelism": f"image shards x{world}, no data-path collective",
                       "algorithmic_bytes_per_image": int(total_alg // B)},
            "roofline": roofline,
            "self_check": self_check,
        }
        if ev_stats is not None:
            out["hip_events"] = ev_stats
        if lat_b1 is not None:
            out["latency_b1"] = lat_b1
        if train is not None:
            out["train_step"] = train
        if train16 is not None:
            out["train_step_bf16"] = train16
        if unpruned is not None:
            out["train_step_unpruned_net"] = unpruned
        if csf is not None:
            out["csf_res2net"] = csf
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(man)
        print(json.dumps(out))
    D.finalize()


def latency_b1(model, dev, n=200):
    """The reference's actual inference loop (CSNet/test.py:87-96): ONE picture at a time -- device resize + normalise
    (csn_resize_normalize_nchw), the replayed B = 1 forward, sigmoid + resize back + uint8 (csn_saliency_resize_u8) --
    timed per picture with HIP events on the launch stream."""
    import torch
    from sod100k_amd import _native as N
    lib = getattr(model, "_lib", None) or N.load()
    h, w = 300, 400
    pic = torch.rand(1, h, w, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    x1 = torch.empty(1, 3, 224, 224, device=dev)                 # fixed buffers: the forward replays its hipGraph
    y1 = torch.empty(1, 1, 224, 224, device=dev)
    u8 = torch.empty(h, w, dtype=torch.uint8, device=dev)
    eng = model.engine_for(x1)
    eng.refresh(model._arena.flat)
    st = torch.cuda.current_stream(dev).cuda_stream

    def one():
        N.check(lib, lib.csn_resize_normalize_nchw(pic.data_ptr(), x1.data_ptr(), 1, h, w, 224, 224, st), "csn_resize_normalize_nchw")
        eng.forward(x1, out=y1)
        N.check(lib, lib.csn_saliency_resize_u8(y1.data_ptr(), u8.data_ptr(), 224, 224, h, w, st), "csn_saliency_resize_u8")

    for _ in range(10):
        one()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a_, b_ in evs:
        a_.record(); one(); b_.record()
    torch.cuda.synchronize(dev)
    ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
    eng.profile(x1, iters=2)
    launches = sum(int(v[1]) for v in eng.kernel_stats().values())
    return {"what": "one 300x400 picture: resize+normalise -> B=1 224x224 forward (hipGraph replay) -> sigmoid+resize back+uint8, "
                    "HIP events per picture", "pictures": n, "median_ms": round(statistics.median(ts), 4), "min_ms": round(ts[0], 4),
            "p90_ms": round(ts[int(0.9 * (n - 1))], 4), "kernel_launches_per_forward": launches,
            "images_per_sec_at_median": round(1e3 / statistics.median(ts), 1)}


def csf_point(dev, batch, steps):
    """BASELINE config 5: CSF+Res2Net-50 eval forward, batch x 3x352x352 fp32, random-init weights.  The decoder head
    is the HIP implicit-GEMM path (include/csf_hip.h); the backbone's convolutions are issued through PyTorch-ROCm /
    MIOpen, its BatchNorm + residual + ReLU passes through csf_bn_act."""
    import torch
    from sod100k_amd.networks import csf_res2net as R
    torch.cuda.empty_cache()
    net = R.build_model().to(dev).eval()
    x = torch.randn(batch, 3, 352, 352, generator=torch.Generator().manual_seed(5)).to(dev)

    def timed(fn, n):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    with torch.no_grad():
        feats = [f.contiguous() for f in net.base(x)]
        ms_head = timed(lambda: net.head_forward(feats, x.shape[2:]), steps)
        ms_all = timed(lambda: net(x), steps)
    eng = list(net._engines.values())[-1]
    tf = 2.0 * eng.macs / (ms_head * 1e-3) / 1e12
    peak = 157.3                               # fp32 matrix-core peak, MI355X_MICROARCH.md (dense, no sparsity)
    return {"workload": f"CSF+Res2Net-50 eval forward, batch {batch} x 3x352x352 fp32, random-init weights",
            "value": round(batch / (ms_all * 1e-3), 1), "unit": "images/sec", "ms_per_step": round(ms_all, 3),
            "ms_head_hip": round(ms_head, 3), "ms_backbone_miopen": round(ms_all - ms_head, 3),
            "head_roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                              "frac": round(tf / peak, 4), "kernel": "csf_gemm3_kernel: fp32 operands as three bfloat16 parts on v_mfma_f32_32x32x16_bf16 (+ combine / GroupNorm passes)",
                              "flops_per_step": int(2 * eng.macs)}}


def eng_sub(eng, B):
    return int(eng.activation(1).shape[0])


if __name__ == "__main__":
    main()
