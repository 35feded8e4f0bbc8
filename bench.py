#!/usr/bin/env python
"""bench.py -- LoANs localizer + assessor joint training step on N MI355X of one node.

One "step" = one ``SheepAssessor.update_core`` (reference sheep/sheep_updater.py:26-68):
assessor forward on the labelled batch, localizer forward, STN crop, assessor forward on the
crops, localizer-chain backward + Adam-AMSGrad, assessor-chain backward + Adam-AMSGrad.

Workload (BASELINE.json): configs[1] shape per GPU -- batch 256 x 3 x 224 x 224 fp32 (the full joint step, a
superset of "localizer forward+backward only").  For N>1 every rank keeps that per-GPU batch (weak scaling, the
per-GPU work is identical at every N) and gradients are all-reduced over RCCL.  With N>1 and no --batch the per-GPU batch
is 128 = configs[3] as BASELINE.json states it (global 1024 at N=8); N=1 stays at 256 = configs[1], and its line carries a
"configs[3] per GPU" secondary leg (128 x 3 x 224 x 224 on one GPU) -- the like-for-like denominator of the N>1 values.
Synthetic paste-and-crop frames, random-init weights; inputs are resident in HBM before timing.

At N = 1 with the default workload the same process then measures the other single-GPU configurations of BASELINE.json as
short SECONDARY legs -- configs[2] (bf16 storage arm, 128 x 3 x 512 x 512) and one GPU's share of configs[4] (ResNet-50
localizer, bf16, 64 x 3 x 512 x 512) -- and reports them under "secondary" in the same line, each with its own roofline; the
primary keys describe the fp32 leg alone (its timed region is closed before a secondary leg starts).

Prints ONE JSON line (rank 0), COMPACT (a few KB: `compact_line`), with the contract keys plus
  roofline     : the ResNet-18 conv-forward MFMA roofline, measured live with HIP events
                 around the forward implicit-GEMM launches (21 convs) of every timed step
                 (minus the time a bracket of two events takes with nothing in it)
  cpu_baseline : the CPU oracle ("port": NumPy restatement of the Chainer graph) timed on
                 the host cores on a bounded sample (B=8) of the same workload (N=1 only)
  secondary    : per leg value / ms_per_step / dtype / baseline_config / roofline_frac / binding_frac only.
Everything else (per-layer binding tables, per-class tables, allocator diagnostics, the full secondary legs) goes to
--detail-file (default bench_detail.json beside this script; round 5's line carried all of it, grew to 26.5 KB and the
driver could not parse it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _load_launcher():
    """loans_amd/launch.py by path: importing the package would pull in torch, and the parent of an N-GPU run must stay a
    plain process that never touches the GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_loans_launch', os.path.join(ROOT, 'loans_amd', 'launch.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    # `python bench.py --gpus N` (N > 1) outside a launcher: fork the N ranks (python -m torch.distributed.run ... bench.py
    # <same args>) as a child, relay its output -- rank 0's single JSON line -- and exit with its code.  Does not return then.
    _load_launcher().launch_if_parent(os.path.abspath(__file__))

import numpy as np          # noqa: E402
import torch                # noqa: E402

CONV_FWD_FLOP_PER_IMAGE_224 = 4166615040        # 21 conv contractions, 2 FLOP per MAC
FP32_MFMA_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0                  # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparsity figure)
HBM_ACHIEVABLE_TBS = 6.3                        # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s spec
HBM_READ_TBS, HBM_WRITE_TBS = 6.5, 4.6          # measured on this chip by tools/chip_peaks (profiles/r2_chip_peaks.txt): plain read / fill kernels


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: 256 = configs[1] at N=1, 128 = configs[3] at N>1)')
    ap.add_argument('--image-size', type=int, default=224)
    ap.add_argument('--target-size', type=int, default=75)
    ap.add_argument('--resnet50', action='store_true', help='Resnet50SheepLocalizer backbone (BASELINE configs[4] architecture, fp32)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="arithmetic of the conv contractions: exact fp32 MFMA (parity path) or bf16 MFMA with fp32 accumulate")
    ap.add_argument('--storage', default=None, choices=['f32', 'bf16'],
                    help="storage of the localizer's stage activations / gradients (default: bf16 with --dtype bf16, else f32)")
    ap.add_argument('--graph', action='store_true', help='capture the step into a hipGraph after warm-up: host offload only -- the replay is slower than the eager step (DESIGN 7e)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--tune-file', default=None,
                    help="kernel-tile table: read if the file exists (the run then launches exactly those tiles -- use for the "
                         "rocprofv3 passes of a command), else written by rank 0 after the run")
    ap.add_argument('--traffic-file', default=None,
                    help="PMC traffic summary (tools/pmc_traffic.py) of THIS command for roofline.traffic; default: "
                         "profiles/<tag>_<b256|cfg3|r50>_conv_fwd_hbm_traffic.json of the newest committed round for those workloads")
    ap.add_argument('--no-secondary', action='store_true',
                    help="only the primary leg (the rocprofv3 passes of tools/profile_round.sh profile one workload per command)")
    ap.add_argument('--secondary', default='cfg3,r50,b128,b16',
                    help="secondary legs run after the default primary workload at N = 1: cfg3 = configs[2] (bf16, 128 x 3 x 512 x 512), "
                         "r50 = one GPU's share of configs[4] (ResNet-50 localizer, bf16, 64 x 3 x 512 x 512), b128 = one GPU's share of configs[3] "
                         "(fp32, 128 x 3 x 224 x 224), b16 = the reference's "
                         "own default batch (-b 16, 224 x 224, fp32: the launch-bound regime), eager and as a hipGraph")
    ap.add_argument('--secondary-steps', type=int, default=10)
    ap.add_argument('--secondary-warmup', type=int, default=3)
    ap.add_argument('--secondary-shape', default=None,
                    help="B,HW: shrink the secondary legs and run them behind any primary workload (schema tests on tiny shapes)")
    ap.add_argument('--detail-file', default=os.path.join(ROOT, 'bench_detail.json'),
                    help="where rank 0 writes the FULL result object (per-layer / per-class tables, full secondary legs); the line on "
                         "stdout is its compact form")
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-iters', type=int, default=4)
    return ap.parse_args()


def cpu_baseline(args, hw, crop):
    """Oracle step timed on the host cores (kind "port")."""
    from oracle import model as M
    from loans_amd.datasets import synthetic
    B = args.cpu_batch
    rng = np.random.RandomState(0)
    lp = M.init_localizer_params(rng, predictor_w_std=1e-3)
    dp = M.init_assessor_params(rng, (crop, crop))
    frames = synthetic.make_frames(100, B, hw, hw)
    real, labels = synthetic.make_assessor_batch(101, B, crop, crop)
    og, od = M.AdamAMSGrad(lp), M.AdamAMSGrad(dp)
    times = []
    for i in range(1 + args.cpu_iters):
        t0 = time.perf_counter()
        M.update_core(lp, dp, og, od, frames, real, labels, (crop, crop), rng=np.random.RandomState(0))
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    try:            # the threads NumPy's BLAS actually runs the im2col GEMMs on (the rest of the oracle is one thread)
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info() if p.get('user_api') == 'blas'] or [1])
    except Exception:
        cores = 1
    return {"value": round(B / t, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d joint steps of batch %d x 3 x %d x %d (+%d x 3 x %d x %d crops), NumPy fp32 oracle, "
                      "median after 1 warm-up" % (args.cpu_iters, B, hw, hw, B, crop, crop),
            "s_per_step": round(t, 3)}


def dry_run(args):
    """LOANS_BENCH_DRY=1: the launcher / rendezvous / collective plumbing of an N-rank run WITHOUT the kernels, for hosts
    with no GPU (tests/test_launch_cpu.py: `python bench.py --gpus 2` end to end over gloo).  A "step" all-reduces a
    gradient-arena-sized buffer through the same Communicator the real step uses; the line it prints carries
    value = null and says so in `data` -- it is never a measurement."""
    from loans_amd import parallel
    comm = parallel.init_from_env()
    if comm.size != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, comm.size))
    if os.environ.get('LOANS_BENCH_DRY_FAIL_RANK') == str(comm.rank):
        raise SystemExit(7)                                # the launcher must propagate a rank's failure

    class Arena:
        numel = 1 << 18
        grad = torch.full((1 << 18,), float(comm.rank + 1))
    arena = Arena()
    for _ in range(args.warmup):
        comm.allreduce_grad(arena)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        comm.allreduce_grad(arena)
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    expect = float(sum(range(1, comm.size + 1))) if comm.size > 1 else 1.0
    total = expect
    for _ in range(args.warmup + args.steps - 1):
        total *= comm.size if comm.size > 1 else 1
    assert comm.size == 1 or abs(float(arena.grad[0]) - total) <= 1e-6 * total, (float(arena.grad[0]), total)
    if comm.rank == 0:
        print(json.dumps({"metric": "localizer+assessor train images/sec", "value": None, "unit": "images/s",
                          "n_gpus": comm.size, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": args.dtype,
                          "data": "dry-run: launcher / rendezvous / gradient all-reduce plumbing only, no kernels, not a measurement",
                          "config": {"workload": "dry-run", "world_size": comm.size, "dist_backend": comm.backend,
                                     "parallelism": "dp%d" % comm.size}}), flush=True)
    parallel.shutdown()


PROFILE_TAGS = ('r6', 'r5', 'r4', 'r3', 'r2')          # newest first: the committed rocprofv3 evidence a default workload is tied to

# The workloads with committed evidence under profiles/ (<tag>_<name>_tune.json = the tile table the timed run, the kernel trace
# and the PMC passes of tools/profile_round.sh all ran on; <tag>_<name>_conv_fwd_hbm_traffic.json = the PMC bytes)
STD_WORKLOADS = {(224, 256, 'f32', 'f32', 75, False): 'b256',        # BASELINE configs[1] shape, the primary leg
                 (224, 128, 'f32', 'f32', 75, False): 'b128',        # BASELINE configs[3], one GPU's share (128 of 1024)
                 (512, 128, 'bf16', 'bf16', 75, False): 'cfg3',     # BASELINE configs[2]
                 (512, 64, 'bf16', 'bf16', 75, True): 'r50'}        # BASELINE configs[4], one GPU's share (64 of 512)


def _profile_file(name, kind):
    for tag in PROFILE_TAGS:
        tune = os.path.join(ROOT, 'profiles', '%s_%s_tune.json' % (tag, name))
        if os.path.exists(tune):
            path = os.path.join(ROOT, 'profiles', '%s_%s_%s.json' % (tag, name, kind))
            return path if os.path.exists(path) else None
    return None


def _reference_ms(w):
    """ms_per_step of the committed bench line of this workload (newest profiles/<round>_<name>_bench.json), None if there is none"""
    name = STD_WORKLOADS.get((w.image_size, w.batch, w.dtype, w.storage, w.target_size, w.resnet50))
    path = _profile_file(name, 'bench') if name else None
    try:
        return float(json.load(open(path))["ms_per_step"]) if path else None
    except (OSError, ValueError, KeyError):
        return None


def workload_of(args, **over):
    """the knobs of one leg: the command line's, or a secondary leg's overrides of them"""
    from types import SimpleNamespace
    # no --batch: configs[1] (256) on one GPU, configs[3] as stated (128 per GPU, global 1024 at N = 8) on several
    w = SimpleNamespace(image_size=args.image_size, batch=args.batch or (256 if args.gpus == 1 else 128), dtype=args.dtype, storage=args.storage,
                        resnet50=args.resnet50, target_size=args.target_size, steps=args.steps, warmup=args.warmup,
                        graph=args.graph, tune_file=args.tune_file, traffic_file=args.traffic_file)
    for k, v in over.items():
        setattr(w, k, v)
    w.storage = w.storage or ('bf16' if w.dtype == 'bf16' else 'f32')
    return w


def config_label(w, world):
    hw, B = w.image_size, w.batch
    std18 = hw == 224 and not w.resnet50 and w.dtype == 'f32'
    r50 = hw == 512 and B == 64 and w.resnet50 and w.dtype == 'bf16'
    if world > 1:
        if r50:
            return "configs[4]" if world == 8 else "configs[4] per GPU, data parallel"
        return "configs[3]" if (std18 and B == 128) else ("configs[1] per GPU, data parallel" if (std18 and B == 256) else "custom")
    if std18 and B == 256:
        return "configs[1]"
    if std18 and B == 128:
        return "configs[3] per GPU (128 of the global 1024)"
    if hw == 512 and B == 128 and not w.resnet50 and w.dtype == 'bf16':
        return "configs[2]"
    if r50:
        return "configs[4] per GPU (64 of the global 512)"
    return "custom"


def conv_forward_roofline(w, log, flop_count, ev_overhead_ms, ms_per_step, world):
    """`roofline` of one leg from the HIP events of its timed steps (ops.EVENT_LOG): the localizer's conv-forward launches
    against the MFMA peak (`frac`), against the roofline that binds each launch (`binding`), the PMC bytes of the same
    launches (`traffic`) and every convolution of the step against the same peak over the step's wall time (`whole_step`)."""
    B, hw, steps = w.batch, w.image_size, w.steps
    rows = []
    for x in log:
        x = tuple(x)
        x = (x + (1, 1, (0, 0)))[:7] if len(x) < 7 else x          # (tag, flops, ev0, ev1, launches, convolutions, (read, written) bytes)
        rows.append(x)
    loc = [(tag, flops, max(s.elapsed_time(e) - ev_overhead_ms, 0.0), nl, nc, nb) for tag, flops, s, e, nl, nc, nb in rows if tag == 'fprop_bn']
    raw_ms = sum(s.elapsed_time(e) for tag, flops, s, e, nl, nc, nb in rows if tag == 'fprop_bn')
    tot_ms = sum(x[2] for x in loc)
    tot_flop = sum(x[1] for x in loc)
    n_launch = sum(x[3] for x in loc)          # a LOANS_TILE_SPLIT conv is two launches
    achieved = tot_flop / (tot_ms * 1e-3) / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS if w.dtype == 'bf16' else FP32_MFMA_PEAK_TFLOPS
    if w.dtype != 'bf16':
        kernels = "igemm_kernel + stem7_kernel"
    elif w.resnet50:
        kernels = "igemm16_kernel (1x1 and 3x3) + halo16 kernels + pw16_kernel (res2 / res3 expansions) + stem7_bf16_kernel"
    else:
        kernels = "igemm16_kernel + halo16 / ws8 / wsw kernels + stem7_bf16_kernel"
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None,
                "kernel": "%s (localizer conv forward: %d convs = %d launches per step)"
                          % (kernels, sum(x[4] for x in loc) // steps, n_launch // steps),
                "avg_launch_ms": round(tot_ms / max(n_launch, 1), 4),
                "conv_fwd_ms_per_step": round(tot_ms / steps, 3),
                "conv_fwd_ms_per_step_raw_brackets": round(raw_ms / steps, 3),
                "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2),
                "event_brackets_inside_timed_region": "%d pairs per step (~%.2f ms of ms_per_step)"
                                                      % (len(rows) // steps, len(rows) / steps * ev_overhead_ms),
                "algorithmic_flop_per_step": tot_flop // steps}
    roofline["launches_per_step"] = n_launch // steps
    # The same launches against the roofline that BINDS each of them: a layer cannot run faster than its algorithmic
    # bytes at the achievable HBM rate, nor than its algorithmic FLOP at the MFMA peak.  In fp32 every layer is MFMA-bound
    # (205 FLOP/B against a balance of 25); in bf16 the stem and res2 are HBM-bound, so grading them against
    # the MFMA peak alone would ask the impossible of them.  Two prices for the bytes: everything at the 6.3 TB/s a streaming
    # kernel reaches (`frac`, rounds 1-2), and reads at 6.5 / writes at 4.6 TB/s -- what tools/chip_peaks measured on this chip
    # for plain read / fill kernels (`frac_write_priced`; a conv whose bytes are mostly output is bound by the write rate).
    per = {}
    for tag, flops, ms, nl, nc, nb in loc:
        e = per.setdefault((flops, tuple(nb)), [0, 0.0])
        e[0] += 1
        e[1] += ms
    t_bound = t_bound_rw = t_meas = 0.0
    n_hbm = n_warn = 0
    layers = []
    for (flops, (rd, wr)), (cnt, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        t_f = flops / (peak * 1e12) * 1e3
        t_b = (rd + wr) / (HBM_ACHIEVABLE_TBS * 1e12) * 1e3
        t_rw = rd / (HBM_READ_TBS * 1e12) * 1e3 + wr / (HBM_WRITE_TBS * 1e12) * 1e3
        bound_ms, bound_rw = max(t_f, t_b), max(t_f, t_rw)
        t_bound += bound_ms * cnt
        t_bound_rw += bound_rw * cnt
        t_meas += ms
        n_hbm += cnt if t_b > t_f else 0
        layer = {"gflop": round(flops / 1e9, 2), "mbytes": round((rd + wr) / 1e6, 1), "mbytes_written": round(wr / 1e6, 1),
                 "calls_per_step": cnt // steps, "ms": round(ms / cnt, 4), "bound": "hbm" if t_b > t_f else "mfma",
                 "bound_ms": round(bound_ms, 4), "frac": round(bound_ms * cnt / max(ms, 1e-9), 3),
                 "frac_write_priced": round(bound_rw * cnt / max(ms, 1e-9), 3)}
        # a launch cannot beat its own bound: a fraction above 1 means the byte or FLOP count is wrong, not that the kernel is
        # fast (launches of a few microseconds, where the bracket correction is the measurement, are exempt: tiny test shapes)
        if ms / cnt > 10 * ev_overhead_ms:
            # (the write-priced figure is NOT asserted: 4.6 TB/s is what a plain fill kernel reaches, not a law -- LOANS_TILE_PW's
            # non-temporal row stores under its reads run res2's expansions at 1.00 of it; it is reported as a warning instead)
            assert layer["frac"] <= 1.0, ("conv-forward layer above its roofline", layer)
            if layer["frac_write_priced"] > 1.05:
                layer["warning"] = "above the write-priced bound by more than 5 %: check the byte count"
                n_warn += 1
        layers.append(layer)
    roofline["binding"] = {
        "rule": "per launch max(algorithmic bytes / %.1f TB/s, algorithmic FLOP / %.0f TFLOP/s); algorithmic bytes = the input "
                "pixels the taps address + weights + output, once each" % (HBM_ACHIEVABLE_TBS, peak),
        "frac": round(t_bound / max(t_meas, 1e-9), 4), "bound_ms_per_step": round(t_bound / steps, 3),
        "rule_write_priced": "reads at %.1f TB/s, writes at %.1f TB/s (tools/chip_peaks, profiles/r2_chip_peaks.txt)" % (HBM_READ_TBS, HBM_WRITE_TBS),
        "frac_write_priced": round(t_bound_rw / max(t_meas, 1e-9), 4),
        "hbm_bound_launches_per_step": n_hbm // steps, "layers_above_write_priced_bound": n_warn, "layers": layers}
    # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, separate
    # rocprofv3 runs of this same command on the SAME tile table, see --tune-file / tools/profile_round.sh).  The file
    # carries the launch count of its own pass: bytes are divided by THAT, and a file whose pass launched other
    # kernels than this run does not describe it -- traffic stays null then.
    tpath = w.traffic_file
    name = STD_WORKLOADS.get((hw, B, w.dtype, w.storage, w.target_size, w.resnet50))
    if tpath is None and world == 1 and name:
        tpath = _profile_file(name, 'conv_fwd_hbm_traffic')
    if tpath and os.path.exists(tpath):
        tf = json.load(open(tpath))
        if int(tf["launches"]) == n_launch // steps:
            roofline["traffic"] = round(tf["total_bytes_per_step"] / tf["launches"])
            roofline["traffic_unit"] = "HBM bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE over the %d conv-forward launches of " \
                                       "one step, %s)" % (tf["launches"], os.path.relpath(tpath, ROOT))
            roofline["traffic_bytes_per_step"] = round(tf["total_bytes_per_step"])
            alg = sum((rd + wr) * cnt for (flops, (rd, wr)), (cnt, ms) in per.items()) // steps
            roofline["algorithmic_bytes_per_step"] = alg
            roofline["traffic_over_algorithmic"] = round(tf["total_bytes_per_step"] / max(alg, 1), 3)
        else:
            roofline["traffic_note"] = "%s was taken on %d launches per step, this run has %d: not comparable" % (
                os.path.relpath(tpath, ROOT), tf["launches"], n_launch // steps)
    if hw == 224 and not w.resnet50:
        assert tot_flop // steps == B * CONV_FWD_FLOP_PER_IMAGE_224, (tot_flop // steps, B)
    # the whole step against the same peak: algorithmic FLOP of EVERY convolution launch (localizer and assessor;
    # forward, data gradient, weight gradient) over the step's wall time -- what the three streams together sustain
    if flop_count:
        step_flop = sum(flop_count.values()) // steps
        roofline["whole_step"] = {
            "algorithmic_flop_per_step": step_flop,
            "by_kind": {k: v // steps for k, v in sorted(flop_count.items())},
            "achieved": round(step_flop / (ms_per_step * 1e-3) / 1e12, 2),
            "frac": round(step_flop / (ms_per_step * 1e-3) / 1e12 / peak, 4)}
    return roofline



def whole_step_binding(w, class_count, ms_per_step, world):
    """`roofline.whole_step.binding`: the WHOLE step priced like the conv forward -- per kernel class the algorithmic FLOP and bytes
    of one step (counted live by loans_amd/ops.py: CLASS_COUNT) against max(bytes / 6.3 TB/s, FLOP / MFMA peak), beside the kernel
    time of that class in the committed rocprofv3 trace of this workload on this tile table (profiles/<round>_<name>_class_times.json,
    tools/class_times.py; null without one).  Two bounds for the step: `bound_ms_sum` = every class at its own roofline, one
    after the other (no credit for overlap between streams); `bound_ms_machine` = max(all bytes / HBM rate, all FLOP / peak)."""
    steps = w.steps
    peak = (BF16_MFMA_PEAK_TFLOPS if w.dtype == 'bf16' else FP32_MFMA_PEAK_TFLOPS) * 1e12
    bw = HBM_ACHIEVABLE_TBS * 1e12
    name = STD_WORKLOADS.get((w.image_size, w.batch, w.dtype, w.storage, w.target_size, w.resnet50))
    path = _profile_file(name, 'class_times') if (name and world == 1 and not w.graph) else None
    measured = json.load(open(path))["kernel_ms"] if path else {}
    classes, tot_f, tot_b, bound_sum = {}, 0, 0, 0.0
    for cls, (flop, rd, wr, t_launch) in sorted(class_count.items()):
        flop, rd, wr = flop // steps, rd // steps, wr // steps
        t_f, t_b = flop / peak * 1e3, (rd + wr) / bw * 1e3
        b = t_launch / steps * 1e3          # per LAUNCH max(bytes / HBM rate, FLOP / peak), summed: >= max of the class's sums
        m = measured.get(cls)
        classes[cls] = {"gflop": round(flop / 1e9, 1), "mbytes": round((rd + wr) / 1e6, 1), "bound": "mfma" if t_f >= t_b else "hbm",
                        "bound_ms": round(b, 4), "kernel_ms": m, "frac": round(b / m, 3) if m else None,
                        "gap_ms": round(m - b, 3) if m else None}
        tot_f, tot_b, bound_sum = tot_f + flop, tot_b + rd + wr, bound_sum + b
    machine = max(tot_f / peak, tot_b / bw) * 1e3
    out = {"rule": "per launch max(algorithmic bytes / %.1f TB/s, algorithmic FLOP / %.0f TFLOP/s), summed per class; kernel_ms from %s"
                   % (HBM_ACHIEVABLE_TBS, peak / 1e12, os.path.relpath(path, ROOT) if path else "no committed trace of this workload"),
           "classes": classes, "bound_ms_sum": round(bound_sum, 3), "bound_ms_machine": round(machine, 3),
           "gbytes_per_step": round(tot_b / 1e9, 2), "tflop_per_step": round(tot_f / 1e12, 3),
           "binding_frac": round(bound_sum / ms_per_step, 4), "machine_frac": round(machine / ms_per_step, 4)}
    if measured:
        gaps = sorted(((c, v["gap_ms"]) for c, v in classes.items() if v["gap_ms"] is not None), key=lambda kv: -kv[1])
        out["largest_gaps"] = [{"class": c, "gap_ms": g} for c, g in gaps[:3]]
        out["kernel_ms_sum"] = round(sum(v for v in measured.values()), 3)
    return out

def run_workload(w, comm, local_rank, retune):
    """One leg: build the models of workload `w`, W warm-up steps, exactly K timed steps between barrier + synchronize on
    both sides, max over ranks.  Returns (rank 0) the dict of that leg: value, ms_per_step, config, roofline."""
    import loans_amd
    from loans_amd import ops, parallel
    from loans_amd.datasets import synthetic
    from loans_amd.runtime import training

    world, rank = comm.size, comm.rank
    dev = torch.device('cuda', local_rank)
    B, hw, crop = w.batch, w.image_size, w.target_size
    name = STD_WORKLOADS.get((hw, B, w.dtype, w.storage, crop, w.resnet50))
    tune_loaded, tune_file = 0, w.tune_file
    if tune_file is None and not retune and name:
        # a workload whose rocprofv3 evidence is committed under profiles/ runs on the tile table those passes used, so that
        # the timed run, the trace and the PMC counters describe the SAME launches (LOANS_BENCH_RETUNE=1: tune afresh)
        tune_file = _profile_file(name, 'tune')
    if tune_file and os.path.exists(tune_file):
        tune_loaded = ops.load_tune_table(tune_file)

    # ---- synthetic inputs, resident in HBM ----
    pool = 32
    frames = synthetic.make_frames(1000 + rank, pool, hw, hw)
    real, labels = synthetic.make_assessor_batch(2000 + rank, pool, crop, crop)
    reps = (B + pool - 1) // pool
    frames_d = torch.from_numpy(np.tile(frames, (reps, 1, 1, 1))[:B]).to(dev)
    real_d = torch.from_numpy(np.tile(real, (reps, 1, 1, 1))[:B]).to(dev)
    labels_d = torch.from_numpy(np.tile(labels, (reps, 1))[:B]).to(dev)

    # ---- models (random init; param_predictor.W seeded non-zero so the backbone gets gradients) ----
    np.random.seed(1234)
    localizer = (loans_amd.Resnet50SheepLocalizer if w.resnet50 else loans_amd.SheepLocalizer)((crop, crop))
    localizer.param_predictor.W.set_logical(
        (1e-3 * np.random.standard_normal(localizer.param_predictor.W.logical_shape)).astype(np.float32))
    discriminator = loans_amd.ResnetAssessor()
    localizer.set_precision(w.dtype, w.storage)       # the arithmetic is a property of the models, not of the process
    discriminator.set_precision(w.dtype, w.storage)
    with loans_amd.using_config('enable_backprop', False):
        discriminator(real_d[:2])                     # materialise the lazy l4, build the arenas
    localizer.finalize(dev)
    comm.bcast_data(localizer)
    comm.bcast_data(discriminator)

    opt_gen = parallel.create_multi_node_optimizer(loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(localizer), comm)
    opt_dis = parallel.create_multi_node_optimizer(loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(discriminator), comm)
    updater = loans_amd.SheepAssessor(
        models=[localizer, discriminator],
        iterator={'main': training.DeviceBatchIterator([frames_d]),
                  'real': training.DeviceBatchIterator([(real_d, labels_d)])},
        optimizer={'opt_gen': opt_gen, 'opt_dis': opt_dis},
        converter=training.identity_converter, device=local_rank, comm=comm, use_graph=w.graph)

    # kernel-tile autotuning and lazily created links happen on the first step a shape is seen; with --warmup 0 that
    # one-off initialisation would land in the timed region, so it gets a step of its own (reported as init_steps)
    init_steps = 1 if w.warmup == 0 else 0
    for _ in range(init_steps + w.warmup):
        updater.update()

    # ---- timed region: exactly K steps between barrier + synchronize ----
    # HIP events around the conv-forward launches (not under a hipGraph: a replay runs no host code, and events recorded while
    # the step is being captured cannot be read)
    ops.EVENT_LOG = [] if (rank == 0 and not w.graph) else None
    ops.FLOP_COUNT = {} if rank == 0 else None         # algorithmic FLOP of every convolution launch of the timed steps
    ops.CLASS_COUNT = {} if rank == 0 else None        # algorithmic FLOP / bytes of EVERY launch, per kernel class
    comm.barrier()
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_stats(dev)
    arena = ops.step_arena_state(dev)
    t0 = time.perf_counter()
    for _ in range(w.steps):
        updater.update()
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    mem1 = torch.cuda.memory_stats(dev)
    # diagnostics only (measurement hygiene): device allocations / allocator retries INSIDE the timed region -- a
    # steady-state step should make none (everything comes from torch's cache); a hipMalloc / hipFree there synchronises the device
    allocator = {k: int(mem1.get(k, 0) - mem0.get(k, 0)) for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries')}
    # the step's activation workspace (ops._StepArena): its size, what one step takes from it, requests it could not serve
    arena1 = ops.step_arena_state(dev)
    allocator['step_arena'] = {'mbytes': round(arena1['bytes'] / 1e6, 1), 'mbytes_per_step': round(arena1['used'] / 1e6, 1),
                               'requests_not_served': arena1['misses'] - arena['misses']}
    log, ops.EVENT_LOG = ops.EVENT_LOG, None
    flop_count, ops.FLOP_COUNT = ops.FLOP_COUNT, None
    class_count, ops.CLASS_COUNT = ops.CLASS_COUNT, None

    # what the HOST needs to enqueue one step (after the timed region, not part of it): with the GPU idle at the start of
    # each call, update() returns as soon as its last launch is queued.  A value near ms_per_step means the step is launch-bound
    host_ms = []
    for _ in range(3):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        updater.update()
        host_ms.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    host_enqueue_ms = float(np.median(host_ms))

    # what a bracket of two HIP events measures with NOTHING between them: the events are packets of their own in the
    # queue, and that time is not the kernel's (rocprofv3's per-kernel durations do not contain it either)
    ev_overhead_ms = 0.0
    if rank == 0 and log:
        nulls = []
        for _ in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            e1.synchronize()
            nulls.append(e0.elapsed_time(e1))
        ev_overhead_ms = float(np.median(nulls))

    if rank != 0:
        return None
    if w.tune_file and not tune_loaded:
        ops.save_tune_table(w.tune_file)
    ms_per_step = elapsed / w.steps * 1e3
    value = B * world * w.steps / elapsed
    roofline = conv_forward_roofline(w, log, flop_count, ev_overhead_ms, ms_per_step, world) if log else None
    if roofline is not None and class_count:
        roofline.setdefault("whole_step", {})["binding"] = whole_step_binding(w, class_count, ms_per_step, world)
    backbone = "ResNet-50" if w.resnet50 else "ResNet-18"
    return {
        "value": round(value, 2), "unit": "images/s", "steps": w.steps, "warmup": w.warmup, "ms_per_step": round(ms_per_step, 3),
        "dtype": w.dtype,
        "config": {"workload": "LoANs joint step: %s localizer + STN crop + assessor, fwd+bwd+2xAdam-AMSGrad" % backbone,
                   "per_gpu_batch": B, "global_batch": B * world, "frame": "3x%dx%d" % (hw, hw),
                   "crop": "3x%dx%d" % (crop, crop), "parallelism": "dp%d" % world, "world_size": world, "dist_backend": comm.backend,
                   "baseline_config": config_label(w, world), "hip_graph": bool(w.graph), "init_steps": init_steps,
                   "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
                   "allocator_in_timed_region": allocator,
                   "activation_storage": w.storage,
                   "tune_table": ("read %d shapes from %s" % (tune_loaded, os.path.relpath(tune_file, ROOT))) if tune_loaded
                   else "autotuned in this run"},
        "roofline": roofline,
    }


def _pick(d, keys):
    return {k: d[k] for k in keys if d and k in d}


def compact_roofline(r):
    """the `roofline` object of the stdout line: the contract keys and the few figures a reader checks them with; the tables stay
    in the detail file"""
    if not r:
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches_per_step",
                    "conv_fwd_ms_per_step", "algorithmic_flop_per_step", "traffic_over_algorithmic"))
    ws = r.get("whole_step") or {}
    b = ws.get("binding") or {}
    out["binding_frac"] = (r.get("binding") or {}).get("frac")
    out["whole_step"] = {"frac": ws.get("frac"), "binding_frac": b.get("binding_frac"), "machine_frac": b.get("machine_frac")}
    return out


def compact_line(out, detail_file=None):
    """The ONE stdout line.  Round 5 printed the whole result (26.5 KB with three per-layer tables) and the driver, which keeps a
    bounded tail of stdout, could not parse it: the line is now the contract keys, `config`, a reduced `roofline`, `cpu_baseline`
    and five numbers per secondary leg (tests/test_host_cpu.py holds it under 4 KB on a full-size payload)."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["config"] = _pick(out.get("config"), ("workload", "per_gpu_batch", "global_batch", "frame", "crop", "parallelism",
                                               "world_size", "dist_backend", "baseline_config", "hip_graph", "activation_storage",
                                               "tune_table"))
    line["roofline"] = compact_roofline(out.get("roofline"))
    if "cpu_baseline" in out:
        line["cpu_baseline"] = out["cpu_baseline"]
    if out.get("secondary"):
        sec = {}
        for label, leg in out["secondary"].items():
            r = leg.get("roofline") or {}
            b = (r.get("whole_step") or {}).get("binding") or {}
            e = {"value": leg["value"], "ms_per_step": leg["ms_per_step"], "dtype": leg["dtype"],
                 "baseline_config": leg["config"]["baseline_config"], "per_gpu_batch": leg["config"]["per_gpu_batch"],
                 "frame": leg["config"]["frame"], "roofline_frac": r.get("frac"), "conv_fwd_ms_per_step": r.get("conv_fwd_ms_per_step"),
                 "binding_frac": b.get("binding_frac")}
            for k in ("graph_over_eager", "remeasured"):
                if k in leg:
                    e[k] = leg[k] if k != "remeasured" else leg[k]["ms_per_step"]
            sec[label] = e
        line["secondary"] = sec
    if detail_file:
        line["detail"] = os.path.relpath(detail_file, ROOT) if os.path.abspath(detail_file).startswith(ROOT) else detail_file
    return line


B16_EAGER, B16_GRAPH = "reference default (-b 16, 224 x 224), eager", "reference default (-b 16, 224 x 224), hipGraph"


def secondary_legs(args):
    """The other single-GPU configurations of BASELINE.json, measured in the same process after the primary leg so that the
    driver's one bench line carries them: configs[2] (bf16 joint step, 128 x 3 x 512 x 512) and one GPU's share of configs[4]
    (ResNet-50 localizer, bf16, 64 x 3 x 512 x 512).  --secondary-shape B,HW shrinks both (schema tests on tiny shapes)."""
    legs = [("configs[2]", 'cfg3', dict(image_size=512, batch=128, dtype='bf16', storage='bf16', resnet50=False)),
            ("configs[4] per GPU", 'r50', dict(image_size=512, batch=64, dtype='bf16', storage='bf16', resnet50=True)),
            # configs[3] as BASELINE.json states it is 128 frames per GPU: what ONE GPU does with that shard (the N>1 default)
            ("configs[3] per GPU", 'b128', dict(image_size=224, batch=128, dtype='f32', storage='f32', resnet50=False)),
            # the reference's own defaults (train_sheep_localizer.py:56-58: -b 16, 224 x 224, crop 75 x 75): ~600 launches of a
            # few microseconds each -- the regime a captured step is for
            (B16_EAGER, 'b16', dict(image_size=224, batch=16, dtype='f32', storage='f32', resnet50=False, graph=False)),
            (B16_GRAPH, 'b16', dict(image_size=224, batch=16, dtype='f32', storage='f32', resnet50=False, graph=True))]
    want = [s for s in args.secondary.split(',') if s]
    legs = [(label, over) for label, key, over in legs if key in want]
    for label, over in legs:
        over.setdefault('graph', False)
        over.update(steps=args.secondary_steps, warmup=args.secondary_warmup, tune_file=None, traffic_file=None)
        if args.secondary_shape:
            b, hw = (int(v) for v in args.secondary_shape.split(','))
            over.update(batch=b, image_size=hw)
        elif label in (B16_EAGER, B16_GRAPH):
            over.update(steps=max(30, args.secondary_steps))           # 6 ms steps: 30 of them for a stable figure
    return legs


def main():
    args = parse()
    if os.environ.get('LOANS_BENCH_DRY') == '1':
        return dry_run(args)
    import gc
    from loans_amd import parallel

    retune = os.environ.get('LOANS_BENCH_RETUNE') == '1'
    comm = parallel.init_from_env()
    world, rank = comm.size, comm.rank
    if world != args.gpus:
        # `python bench.py --gpus N` forks its own ranks (top of this file); this is a rank started with a mismatching count
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    primary_w = workload_of(args)
    primary = run_workload(primary_w, comm, local_rank, retune)
    secondary = {}
    is_default = STD_WORKLOADS.get((primary_w.image_size, primary_w.batch, primary_w.dtype, primary_w.storage,
                                    primary_w.target_size, primary_w.resnet50)) == 'b256' and not primary_w.graph
    if world == 1 and (is_default or args.secondary_shape) and not args.no_secondary:
        for label, over in secondary_legs(args):
            gc.collect()
            torch.cuda.empty_cache()
            leg = run_workload(workload_of(args, **over), comm, local_rank, retune)
            # Measurement hygiene: on this pool one leg of an otherwise normal line now and then comes out 1.4 - 2 x
            # slow (host or device interference on the box; two of nine runs in round 4).  A SECONDARY leg that is more than
            # 1.3 x its committed reference (profiles/<round>_<name>_bench.json) is measured once more, the faster of the two is
            # reported and the line says so.  The primary leg is never repeated: the contract times exactly K steps once.
            ref_ms = _reference_ms(workload_of(args, **over))
            if rank == 0 and ref_ms and leg["ms_per_step"] > 1.3 * ref_ms:
                # (no selection -- the FIRST measurement stays the leg's figure, the repeat is attached beside it)
                again = run_workload(workload_of(args, **over), comm, local_rank, retune)
                leg["remeasured"] = {"why": "this measurement was more than 1.3 x the committed %.3f ms/step; measured once more, "
                                            "both reported, the first one kept" % ref_ms,
                                     "ms_per_step": [leg["ms_per_step"], again["ms_per_step"]],
                                     "value": [leg["value"], again["value"]]}
            leg["metric"] = "localizer+assessor train images/sec"
            secondary[label] = leg
        if B16_EAGER in secondary and B16_GRAPH in secondary and rank == 0:
            secondary[B16_GRAPH]["graph_over_eager"] = round(secondary[B16_GRAPH]["value"] / secondary[B16_EAGER]["value"], 3)
    if rank != 0:
        parallel.shutdown()
        return
    out = {
        "metric": "localizer+assessor train images/sec", "value": primary["value"], "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": primary["config"], "roofline": primary["roofline"],
    }
    if secondary:
        out["secondary"] = secondary
    if world == 1 and not args.no_cpu_baseline and not args.resnet50 and args.dtype == 'f32':
        out["cpu_baseline"] = cpu_baseline(args, args.image_size, args.target_size)
    detail = args.detail_file
    try:
        with open(detail, 'w') as f:
            json.dump(out, f)
    except OSError as err:                      # a read-only tree: the line still goes out
        print('bench.py: could not write %s: %s' % (detail, err), file=sys.stderr)
        detail = None
    print(json.dumps(compact_line(out, detail)), flush=True)
    parallel.shutdown()


if __name__ == '__main__':
    main()
