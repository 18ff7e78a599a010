#!/usr/bin/env python3
"""bench.py -- train seq/s of the Transformer hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]

One "step" = one full training step (forward + CrossEntropyLoss + backward +
clip_grad_norm_(0.5) + SGD-momentum) of BASELINE.json configs[1]
(Transformer d_model=512, 6 layers, 8 heads, dim_feedforward=512, batch=50,
len=48, |src|=3000, |tgt|=202, dropout 0.1) on one batch of synthetic ASL-Phono
token ids that is already resident in HBM (hipGraph replay or plain stream launches,
whichever a short probe during warmup finds faster -- slnlp/launch.py).
N>1: one process per GPU over RCCL (the driver launches the ranks with
torch.distributed.run; run by hand without WORLD_SIZE, ``--gpus N`` spawns the N
rank processes itself before anything touches the GPU).  Two legs per N:

* ``value``: every rank runs its own independent fit (the reference's only
  parallelism is the embarrassingly parallel (candidate x fold) grid, SURVEY.md
  section 8e) -> weak scaling, no data-path collective;
* ``grid``: the other half of BASELINE.json's metric -- folds/hr of ONE bounded
  cross-validated grid search (the same sample at every N -> strong scaling)
  run by ShardedGridSearchCV over the N ranks: dataset broadcast over RCCL,
  dynamic work counter, one all_gather of the scores.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1] -- the config `metric` is quoted on
    "cfg2": dict(E=512, H=8, N=6, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1),
    # configs[0] -- the reference's own CPU-runnable debug case (.vscode/launch.json:26)
    "cfg1": dict(E=128, H=4, N=2, F=256, Vs=3000, Vt=202, B=50, S=48, dropout=0.1),
    "e1024": dict(E=1024, H=8, N=6, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1),
    # configs[4] shape (d_model 1024, 6 layers, batch 256, len 64; F not stated in BASELINE.json -> grid max 512); split-bf16
    # by default, `--precision 8` runs its forward products on the fp8 MFMA (configs[4]'s "fp8 MFMA weights")
    "cfg5": dict(E=1024, H=8, N=6, F=512, Vs=3000, Vt=202, B=256, S=64, dropout=0.1),
    # configs[2] -- EncoderDecoderLSTMAttn hidden=512, 4 layers, batch=50
    "cfg3": dict(rnn="lstm", E=512, Hd=512, N=4, Vs=3000, Vt=202, B=50, S=48, dropout=0.1),
    "cfg3gru": dict(rnn="gru", E=512, Hd=512, N=4, Vs=3000, Vt=202, B=50, S=48, dropout=0.1),
}
BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
MOMENTUM, MAX_NORM, LR = 0.9, 0.5, 0.01


def fwd_flops_per_seq(c):
    """SURVEY.md section 8d contract: 2mnk per GEMM, attention dense, T=1."""
    if "rnn" in c:   # enc-dec RNN: x/h projections of the bi-encoder, key layer, one decoder step, generator
        E, Hd, N, S, V, G = c["E"], c["Hd"], c["N"], c["S"], c["Vt"], 4 if c["rnn"] == "lstm" else 3
        enc = sum(2 * S * 2 * G * Hd * ((E if l == 0 else 2 * Hd) + Hd) for l in range(N))
        dec = sum(2 * G * Hd * ((E + 2 * Hd if l == 0 else Hd) + Hd) for l in range(N))
        return enc + dec + 2 * S * 2 * Hd * Hd + 2 * N * 2 * Hd * Hd + 2 * Hd * Hd + 4 * S * Hd + 4 * S * Hd + 2 * Hd * V
    E, F, N, S, V = c["E"], c["F"], c["N"], c["S"], c["Vt"]
    enc = N * S * (8 * E * E + 4 * S * E + 4 * E * F)
    dec = N * (12 * E * E + 4 * E + 4 * S * E * E + 4 * S * E + 4 * E * F)
    return enc + dec + 2 * E * V


def build_sd(c, seed):
    from slnlp import synth, tf_engine as te
    if "rnn" in c:
        from slnlp import rnn_engine as re_
        cfg = re_.make_config(c["rnn"], c["E"], c["Hd"], c["N"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, 0, c["dropout"],
                              c.get("precision", 3))
        ents, _ = re_.layout(cfg)
        w = synth.make_weights([(n, s) for n, s, _ in ents], seed=seed)
        return cfg, {k: torch.from_numpy(v) for k, v in w.items()}
    cfg = te.make_config(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, c["dropout"],
                         c.get("precision", 3))
    ents, _ = te.layout(cfg)
    w = synth.make_weights([(n, s) for n, s, _ in ents], seed=seed)
    return cfg, {k: torch.from_numpy(v) for k, v in w.items()}


PROFILE_ROUNDS = ("r05", "r04")  # committed profiles the provenance-labelled fields are read from: the newest round that holds the file


def profile_file(name):
    """profiles/<round>_<name> of the newest round that has it (every value read from it is labelled with the path)."""
    for r in PROFILE_ROUNDS:
        f = os.path.join(ROOT, "profiles", f"{r}_{name}")
        if os.path.exists(f):
            return f
    return os.path.join(ROOT, "profiles", f"{PROFILE_ROUNDS[0]}_{name}")


def backward_passes():
    """(wgrad, dgrad) split-bf16 passes the plans use for their plane-GEMM gradient products (slnlp_set_backward_passes)."""
    import ctypes as C
    from slnlp._lib import load
    w, d = C.c_int32(3), C.c_int32(3)
    load().slnlp_get_backward_passes(C.byref(w), C.byref(d))
    return int(w.value), int(d.value)


def in_step_plane_launches(run_eager_steps, steps=12):
    """Time every plane-GEMM group launch INSIDE train steps: `steps` eager steps with the library's launch timer on (two HIP
    events around each launch, on the launch stream).  Returns [(workgroups, jobs, geometry, us)]; the first two steps are warm-up."""
    import ctypes as C
    from slnlp._lib import load, check, TimedLaunch
    run_eager_steps(2)
    torch.cuda.synchronize()
    check(load().slnlp_launch_timer_start(steps * 128), "launch_timer_start")
    run_eager_steps(steps)
    torch.cuda.synchronize()
    buf = (TimedLaunch * (steps * 128))()
    n = load().slnlp_launch_timer_stop(buf, steps * 128)
    return [(int(buf[i].blocks), int(buf[i].njobs), int(buf[i].geometry), float(buf[i].us)) for i in range(max(n, 0))]


def dominant_kernel_roofline(c, precision, dev, workload, in_step=None):
    """The step's dominant kernel: one grouped plane-GEMM launch = data gradient + weight gradient of one [B*S, E] dY against a
    [E, F] weight (FFN / out-proj pair of the encoder backward; 18 of them per cfg2 step plus 6 larger in_proj ones).
    `achieved` / `frac` price the launch WHERE IT RUNS: the average of its launches inside train steps (`in_step`: HIP events
    around each launch on the launch stream, in_step_plane_launches) -- caches as the step leaves them, the neighbours a step
    gives it.  The same launch back to back (warm L2, no neighbours) is reported beside it as the secondary figure."""
    from slnlp import ops
    M, E, F = c["B"] * c["S"], c["E"], c["F"]
    if E % 64 or F % 64:
        return None
    wp, dp = backward_passes() if precision == 3 else (precision, precision)
    g = torch.Generator().manual_seed(0)
    dY, X, W = [torch.randn(*sh, generator=g).to(dev) for sh in ((M, E), (M, F), (E, F))]
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(E, device=dev)
    jw, _ = ops.plane_job(dYp, Xp, M=E, N=F, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=wp)
    jd, _ = ops.plane_job(dYp, Wp, M=M, N=F, K=E, a_kmajor=True, b_kmajor=False, precision=dp)
    cd = lambda a, b: (a + b - 1) // b
    # what the library does with this pair (csrc/gemm_planes.hip: gemm_planes_wd_plan, plane_geo_auto): the weight gradient's K-split,
    # one grouped launch or two, and the tile geometry of the launch
    split, separate, geo_w, geo_d = ops.gemm_wd_plan(jw, jd)
    scr = ops.gemm_wd(jw, jd)
    for _ in range(20):
        ops.gemm_wd(jw, jd, scr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        ops.gemm_wd(jw, jd, scr)
    e1.record()
    e1.synchronize()
    us_b2b = e0.elapsed_time(e1) / n * 1e3
    flops = 2.0 * 2 * M * E * F                       # two GEMMs, 2 m n k each (the split-bf16 MFMA passes are not counted)
    GEO_DIMS = [(64, 64, "64x64"), (128, 128, "128x128 64-k ring"), (128, 128, "128x128 32-k ring"), (256, 256, "256x256 32-k ring")]
    bm, bn, geo = GEO_DIMS[geo_d]
    wgs = cd(E, bm) * cd(F, bn) * split + cd(M, bm) * cd(F, bn)           # workgroups of the (grouped) launch
    blocks = wgs
    # this launch inside train steps: the timer's records with this workgroup count and two jobs
    mine = [us for (b, nj, g_, us) in (in_step or []) if b == wgs and nj == 2 and g_ == geo_d] if not separate else []
    us_step = sum(mine) / len(mine) if mine else None
    us = us_step if us_step else us_b2b
    tf = flops / (us * 1e-6) / 1e12
    prof = rocprof_kernel_times(workload)
    traffic, traffic_src = pmc_kernel_traffic(workload, f"gemm_planes_kernel<{precision}, {geo_d}> x{blocks}")
    return {"kernel": f"gemm_planes_kernel<{precision}, {geo}> dgrad+wgrad group [{M}x{E}]x[{E}x{F}], wgrad split-K {split}, {wgs} workgroups, "
                      f"split-bf16 passes wgrad {wp} / dgrad {dp}",
            "bound": "mfma", "achieved": round(tf, 1), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / BF16_DENSE_PEAK_TFLOPS, 4),
            "population": (f"in-step: {len(mine)} launches inside {len(mine) // 18 if len(mine) >= 18 else '?'}+ eager train steps, HIP events around each launch on the launch stream (live)"
                           if us_step else "back-to-back launches (live; no in-step records)"),
            "us_per_launch": round(us, 2),
            # the two populations under keys of their own (never a silent stand-in for each other): None = not measured in this run
            "us_per_launch_in_step": round(us_step, 2) if us_step else None,
            "achieved_in_step": round(flops / (us_step * 1e-6) / 1e12, 1) if us_step else None,
            "achieved_is": "in_step" if us_step else "back_to_back",
            "us_per_launch_back_to_back": round(us_b2b, 2), "achieved_back_to_back": round(flops / (us_b2b * 1e-6) / 1e12, 1),
            "frac_back_to_back": round(flops / (us_b2b * 1e-6) / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4),
            "us_per_launch_rocprof": prof,
            "traffic": traffic, "traffic_source": traffic_src,
            "held_clock_ghz": held_clock(),
            "mfma_pipe_frac": round((wp + dp) / 2.0 * tf / BF16_DENSE_PEAK_TFLOPS, 4),
            "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": 4.0 * (M * E + M * F + E * F) + 4.0 * (M * F + E * F) + 4.0 * M * F,
            "note": "flops = 2 GEMMs x 2mnk (algorithmic: the split-bf16 MFMA passes are not counted; mfma_pipe_frac = executed passes x frac, "
                    "at the 2.4 GHz the 2.5 PF peak assumes -- the chip holds held_clock_ghz under this kernel); bytes = operand planes (hi+lo) read "
                    "once + fp32 results + result planes; us_per_launch brackets each launch with two HIP events on its stream, which adds the "
                    "dispatch gap a step really pays (rocprofv3's kernel-only stamps of the same command: us_per_launch_rocprof; two events "
                    "around nothing read ~5 us on a busy stream, so the bracket is not corrected); fields with a *_source / source key are read from committed profiles of an earlier run "
                    "of this command, everything else is measured in this run"}


def held_clock():
    """Shader clock the chip holds under the plane GEMM (probe build: s_memtime / s_memrealtime over the K loop after 2 s of
    back-to-back launches on random data), from the committed probe output -- not measurable in the product build."""
    f = profile_file("plane_gemm_clock.txt")
    try:
        vals = {}
        for line in open(f):
            if "held shader clock" in line:
                name = line.split(" tile ")[0].strip()
                tile = line.split(" tile ")[1].split(":")[0].strip()
                p50 = float(line.split("p10/p50/p90")[1].split()[1])
                vals[f"{name} [tile {tile}]"] = p50
        return {"p50_by_launch": vals, "source": os.path.relpath(f, ROOT)} if vals else None
    except Exception:
        return None


FP8_DENSE_PEAK_TFLOPS = 5000.0    # MI355X_MICROARCH.md: ~5 PF dense fp8, the rate of the block-scaled K=128 MFMA
                                  # (v_mfma_scale_f32_16x16x128_f8f6f4) the precision-8 kernel issues since round 3


def large_launch_roofline(precision, dev):
    """The same plane-GEMM kernel on a launch large enough to fill the chip many times over: the dgrad + wgrad group of
    configs[4]'s in_proj ([16384 x 3072] dY, W [3072 x 1024]; 2176 workgroups at the 128 x 128 tile the launch takes by itself).
    What a merged lockstep launch of many fits looks like to the kernel; reported beside the cfg2 launch, never instead of it."""
    from slnlp import ops
    M, Nout, Kin = 16384, 3072, 1024
    wp, dp = backward_passes() if precision == 3 else (precision, precision)
    g = torch.Generator().manual_seed(0)
    dY, X, W = [torch.randn(*sh, generator=g).to(dev) for sh in ((M, Nout), (M, Kin), (Nout, Kin))]
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(Nout, device=dev)
    jw, _ = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=wp)
    jd, _ = ops.plane_job(dYp, Wp, M=M, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=dp)
    split, separate, geo_w, geo_d = ops.gemm_wd_plan(jw, jd)          # the library's own choice for this pair (slnlp_gemm_wd)
    scr = ops.gemm_wd(jw, jd)
    for _ in range(10):
        ops.gemm_wd(jw, jd, scr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for _ in range(n):
        ops.gemm_wd(jw, jd, scr)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = 2.0 * 2 * M * Nout * Kin
    tf = flops / (us * 1e-6) / 1e12
    GEO_NAMES = ["64x64", "128x128 64-k ring", "128x128 32-k ring", "256x256 32-k ring"]
    how = (f"TWO launches (both large: a launch each): wgrad {GEO_NAMES[geo_w]} split-K {split}, dgrad {GEO_NAMES[geo_d]}" if separate
           else f"ONE grouped launch, {GEO_NAMES[geo_d]}, wgrad split-K {split}")
    traffic, traffic_src = pmc_large_launch_traffic()
    return {"kernel": f"gemm_planes_kernel<{precision}> dgrad + wgrad of one dY [{M}x{Nout}]x[{Nout}x{Kin}] (configs[4] in_proj) as the plans launch it: {how}; "
                      f"split-bf16 passes wgrad {wp} / dgrad {dp}",
            "bound": "mfma", "achieved": round(tf, 1), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / BF16_DENSE_PEAK_TFLOPS, 4),
            "population": "back-to-back pairs, HIP events around the loop (live)",
            "traffic": traffic, "traffic_source": traffic_src, "us_per_launch_hip_events": round(us, 1), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": 4.0 * (M * Nout + M * Kin + Nout * Kin) + 4.0 * (M * Kin + Nout * Kin) + 4.0 * M * Kin,
            "mfma_pipe_frac": round((wp + dp) / 2.0 * tf / BF16_DENSE_PEAK_TFLOPS, 3),
            "note": "algorithmic FLOPs (the split-bf16 MFMA passes are not counted: the MFMA pipe executes (wgrad + dgrad passes) / 2 times this)"}


def fp8_kernel_roofline(c, dev):
    """precision 8's own kernel: the largest forward product of an encoder layer, in_proj [B*S, E] x [3E, E]^T, on e4m3
    planes (slnlp_gemm with precision 8), timed back to back with HIP events."""
    from slnlp import ops
    M, E = c["B"] * c["S"], c["E"]
    g = torch.Generator().manual_seed(0)
    X, W = torch.randn(M, E, generator=g).to(dev), (torch.randn(3 * E, E, generator=g) * 0.05).to(dev)
    Xq, _ = ops.quant_rows_fp8(X)
    Wq, sw = ops.quant_rows_fp8(W)
    out = torch.empty(M, 3 * E, device=dev)
    for _ in range(20):
        ops.gemm_fp8(Xq, Wq, M=M, N=3 * E, K=E, col_scale=sw, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        ops.gemm_fp8(Xq, Wq, M=M, N=3 * E, K=E, col_scale=sw, out=out)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = 2.0 * M * 3 * E * E
    tf = flops / (us * 1e-6) / 1e12
    return {"kernel": f"gemm_q8_kernel (block-scaled fp8 MFMA), in_proj [{M}x{E}]x[{3 * E}x{E}]^T",
            "bound": "mfma", "achieved": round(tf, 1), "peak": FP8_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP8_DENSE_PEAK_TFLOPS, 4),
            "traffic": None, "us_per_launch_hip_events": round(us, 2), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": 1.0 * (M * E + 3 * E * E) + 4.0 * M * 3 * E,
            "note": "e4m3 operand planes (1 B/element) read once + fp32 result; v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, "
                    "priced against the 5 PF dense fp8 peak; the fp32 result alone is 4 B per 2K FLOP of output traffic"}


def concurrent_fits(c, precision, dev, ks=(4, 8, 16), steps=30):
    """Aggregate train seq/s of K independent fits of this workload sharing the GPU by advancing in LOCKSTEP through one
    launch sequence (slnlp/lockstep.py: own weights, lr, seed and data per fit; bit-identical to solo fits) -- how
    ShardedGridSearchCV(lockstep=k) runs a work unit.  A single batch-50 fit leaves most CUs idle in its decoder stages."""
    from slnlp import synth, tf_engine as te, rnn_engine as re_
    from slnlp.lockstep import LockstepGroup
    B, S = c["B"], c["S"]
    rows = steps * B
    out = []
    st = torch.cuda.Stream(device=dev)
    total = torch.cuda.get_device_properties(dev).total_memory
    for k in ks:
        engs, data = [], []
        if out and k * per_fit_bytes > 0.35 * total:      # keep the resident fits well inside the device's memory
            continue
        for i in range(k):
            cfg, sd = build_sd(c, seed=101 + i)
            e = (re_.RnnEngine if "rnn" in c else te.TransformerEngine)(cfg, device=dev, seed=101 + i)
            e.load_state(sd)
            e.set_lr(LR)
            Xn, Ln, yn = synth.make_batch(rows, S, c["Vs"], c["Vt"], seed=101 + i)
            engs.append(e)
            data.append((torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev), torch.from_numpy(Ln).to(dev)))
        per_fit_bytes = engs[0].workspace.numel() + 3 * 4 * engs[0].params.numel()
        with torch.cuda.stream(st):
            grp = LockstepGroup(engs)
            grp.set_data(0, [d[0] for d in data], [d[1] for d in data], B, [d[2] for d in data])
            grp.epoch(0, B, True, MOMENTUM, MAX_NORM)          # records the launch program
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            grp.epoch(0, B, True, MOMENTUM, MAX_NORM)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            n = grp.num_launches(0, B, True)
            grp.close()
        out.append({"fits": k, "value": round(k * rows / dt, 1), "unit": "seq/s (aggregate)", "ms_per_lockstep_step": round(dt / steps * 1e3, 3),
                    "launches_per_step": n})
        del engs, data, grp
        torch.cuda.empty_cache()
    return {"mode": "lockstep (one launch sequence for all fits)", "runs": out,
            "note": f"{steps} steps per fit, every fit its own weights / data; the step's kernels carry a fit index (grid.z) or a merged job table"}


# THE folds/hr sample (fixed from round 3 on; rounds 1-2 quoted smaller ones): 96 of config-transformer.yaml's 324 candidates --
# every lr and dropout, two embedding sizes, two hidden sizes, both head counts, two depths -> 16 shapes x 30 (candidate, fold)
# fits = 480 fits of 6 epochs over 4000 samples.  The 3 learning rates x 5 folds of a (shape, dropout) pair can advance in
# lockstep: 32 groups of 15 in four cost classes (E512 N4 / E512 N2 / E128 N4 / E128 N2, 8 each).  ShardedGridSearchCV cuts the
# groups into work units so that every host thread of every GPU has about six (slnlp/grid.py unit_cost_ceiling: 36 units on one
# GPU, 124 of 3-5 fits on eight) and hands them out longest first -- the strong-scaling leg keeps every GPU's four host threads
# busy to the end at 1 / 2 / 4 / 8 GPUs (tests/test_grid_cpu.py simulates that schedule with measured unit times: 0.97 at 8).
GRID_SAMPLE = {"lr": [0.1, 0.01, 0.001], "module__dropout": [0.5, 0.1], "module__embedding_size": [512, 128],
               "module__hidden_size": [512, 256], "module__num_heads": [8, 4], "module__num_layers": [4, 2]}
GRID_CV, GRID_EPOCHS, GRID_SAMPLES = 5, 6, 4000


def grid_factory(ds, dev, max_epochs=GRID_EPOCHS):
    from slnlp.net import NeuralNetClassifier
    return lambda: NeuralNetClassifier(
        module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
        module__embedding_size=512, module__num_heads=4, module__num_layers=2,
        module__hidden_size=256, module__dropout=0.1, criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.01, max_epochs=max_epochs, batch_size=50,
        device=str(dev), gradient_clipping={"gradient_clip_value": 0.5},
        scoring=["neg_log_loss", "accuracy", "precision_weighted", "recall_weighted", "f1_weighted"])   # config-transformer.yaml:9


def grid_folds_per_hour(dev, world, rank, fits_per_gpu=5, lockstep=15):
    """The other half of BASELINE.json's metric: (candidate x fold) fits per hour of the cross-validated grid search,
    on a bounded sample of config-transformer.yaml's grid -- 96 candidates x cv 5 = 480 fits of 6 epochs over 4000
    synthetic samples (batch 50, len 48, |src| 3000, 200 labels) -- run by ShardedGridSearchCV over all `world`
    ranks (rank 0 owns the dataset and broadcasts it; the same sample at every N: strong scaling).  Work unit =
    `lockstep` shape-compatible fits advancing through one launch sequence; `fits_per_gpu` host threads per rank each run
    one unit at a time on a stream of its own (slnlp.net.device_stream), so one unit's host work (estimator construction,
    epoch metrics, scoring) and small launches hide under another's kernels; `scores_crc32` is the checksum of every
    candidate's mean test score -- the same in every run, stream mode and world size."""
    import warnings
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    warnings.filterwarnings("ignore", message="enable_nested_tensor")        # torch modules built only to draw the initial weights
    warnings.filterwarnings("ignore", message="The least populated class")    # 2000 samples over 200 labels
    ds = synthetic_dataset(GRID_SAMPLES, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
    # untimed warm-up on every rank (code objects, allocator pools, plan creation paths): one tiny fit per shape
    warm = ShardedGridSearchCV(grid_factory(ds.truncated(200), dev, 1), {k: GRID_SAMPLE[k] for k in ("module__embedding_size", "module__hidden_size", "module__num_heads")},
                               cv=2, refit=False, device=str(dev), fits_per_gpu=1, schedule="static", lockstep=min(lockstep, 2))
    warm_t0 = time.perf_counter()
    _fit_local(warm, ds.truncated(200))       # every rank warms up on the whole warm-up grid, not on a shard of it
    warm_s = time.perf_counter() - warm_t0
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    gs = ShardedGridSearchCV(grid_factory(ds, dev), GRID_SAMPLE, cv=GRID_CV, refit=False, device=str(dev),
                             fits_per_gpu=fits_per_gpu, lockstep=lockstep).fit(ds if rank == 0 else None)
    dt = time.perf_counter() - t0
    return {"value": round(gs.n_tasks_ / dt * 3600.0, 0), "unit": "folds/hr", "fits": gs.n_tasks_, "seconds": round(dt, 2),
            "fits_per_gpu": fits_per_gpu, "lockstep": lockstep, "work_units": gs.n_units_, "ranks": world, "schedule": gs.schedule,
            "rank_seconds": [round(v, 2) for v in gs.rank_seconds_], "rank_fits": gs.rank_tasks_, "warmup_seconds": round(warm_s, 2),
            "best_index": gs.best_index_, "best_score": round(gs.best_score_, 5),
            # every candidate's mean test score, bit for bit: the same at every N, stream mode and run (fits never influence each other)
            "scores_crc32": "%08x" % zlib.crc32(np.asarray(gs.cv_results_["mean_test_score"], dtype=np.float64).tobytes()),
            "sample": f"{len(gs.cv_results_['params'])} candidates (lr x dropout x embedding_size x hidden_size x num_heads x num_layers of "
                      f"config-transformer.yaml) x cv {GRID_CV}, {GRID_EPOCHS} epochs, {GRID_SAMPLES} samples, 80/20 train/valid split inside each fit, "
                      "the reference's 5 epoch metrics on both; includes the dataset broadcast and the score all_gather"}


def _fit_local(gs, ds):
    """Run a grid search on this rank alone even inside a process group (warm-up)."""
    from slnlp import grid as G
    saved = G._dist
    G._dist = lambda: (None, 0, 1)
    try:
        return gs.fit(ds)
    finally:
        G._dist = saved


def rocprof_kernel_times(workload):
    """The committed rocprofv3 --kernel-trace view of the same kernel (tools/roofline_kernel_stats.py): its back-to-back
    launches and its launches inside train steps, with the file it was read from, or None."""
    f = profile_file(f"bench_{workload}_roofline_kernel.json")
    try:
        e = json.load(open(f))
        return {"back_to_back_avg_us": e["back_to_back"]["avg_us"], "in_step_avg_us": e["in_step"]["avg_us"], "min_us": e["min_us"],
                "source": os.path.relpath(f, ROOT)}
    except Exception:
        return None


def pmc_kernel_traffic(workload, shape):
    """(HBM bytes of one launch of `shape` ("kernel xWORKGROUPS") from the committed PMC passes, the file) or (None, None)."""
    f = profile_file(f"pmc_{workload}_step_traffic.json")
    try:
        e = json.load(open(f))["per_launch"][shape]
        return float(e["fetch_bytes"] + e["write_bytes"]), os.path.relpath(f, ROOT)
    except Exception:
        return None, None


def pmc_large_launch_traffic():
    """(HBM bytes of one launch of the configs[4] in_proj gradient group from the committed PMC passes, the file) or (None, None)."""
    f = profile_file("pmc_large_launch_traffic.json")
    try:
        return float(json.load(open(f))["hbm_bytes"]), os.path.relpath(f, ROOT)
    except Exception:
        return None, None


def pmc_traffic(workload):
    """(HBM bytes per train step from the committed rocprofv3 --pmc passes (tools/pmc_step_traffic.py), the file) or (None, None)."""
    f = profile_file(f"pmc_{workload}_step_traffic.json")
    try:
        return float(json.load(open(f))["hbm_bytes_per_step"]), os.path.relpath(f, ROOT)
    except Exception:
        return None, None


def cpu_parity(c, sd, Xe, ye, Le, logp_gpu):
    """CPU leg, part 1 (the checker): the oracle's eval-mode log-probs for one held-out batch against the HIP path's --
    the "top-1 vs ref" half of BASELINE.json's metric.  With cpu_baseline() below, the only place bench.py touches oracle/."""
    from oracle import rnn_ref, transformer_ref as tr
    if "rnn" in c:
        lo = rnn_ref.forward(sd, Xe, ye, Le, rnn_type=c["rnn"], num_layers=c["N"])
    else:
        lo = tr.forward(sd, Xe, ye, num_heads=c["H"], num_layers=c["N"])
    return {"argmax_agree": float((logp_gpu.argmax(-1) == lo.argmax(-1)).float().mean()),
            "logp_rel_err": float((logp_gpu - lo).abs().max() / lo.abs().max())}


def cpu_baseline(c, sd, X, y, Ln, budget_s=14.0):
    """CPU leg, part 2: the oracle (CPU port of the reference step) timed on this host's cores on a bounded sample.
    Headline variant = BASELINE.md section 3's: all the cores this process may use, dropout ON (fresh Bernoulli
    masks per step, the reference's nn.Dropout cost; 25 % of its CPU step).  Also reported:
    dropout off, and one thread."""
    from oracle import rnn_ref, train_ref, transformer_ref as tr
    # the GPU box gives this process a CPU share (16 cores per GPU), not the whole host
    cores = min(len(os.sched_getaffinity(0)), 16)
    B = c["B"]

    def timed(threads, draw, budget):
        torch.set_num_threads(threads)
        if "rnn" in c:
            fwd = lambda p, X, y, L: rnn_ref.forward(p, X, y, L, rnn_type=c["rnn"], num_layers=c["N"],
                                                     p_drop=c["dropout"] if draw else 0.0, masks="draw" if draw else None)
            frozen = ("model.decoder.pre_output_layer.weight",)
        else:
            fwd = lambda p, X, y, L: tr.forward(p, X, y, num_heads=c["H"], num_layers=c["N"],
                                                p_drop=c["dropout"] if draw else 0.0, masks="draw" if draw else None)
            frozen = ()
        trn = train_ref.Trainer(sd, fwd, pad_tgt=1, lr=LR, momentum=MOMENTUM, max_norm=MAX_NORM, frozen=frozen)
        trn.step(X[:B], y[:B], Ln[:B])  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            i = (n % (X.shape[0] // B)) * B
            trn.step(X[i:i + B], y[i:i + B], Ln[i:i + B])
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= 50:
                break
        return n * B / dt, n, dt

    v, n, dt = timed(cores, True, budget_s)
    v_nodrop, n2, dt2 = timed(cores, False, budget_s * 0.5)
    v_one, n3, dt3 = timed(1, True, budget_s * 0.5)
    torch.set_num_threads(cores)
    return {"value": round(v, 2), "unit": "seq/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of batch {B} ({dt:.1f} s), torch fp32, {cores} threads, dropout {c['dropout']} (masks drawn per step)",
            "dropout_off": {"value": round(v_nodrop, 2), "sample": f"{n2} steps ({dt2:.1f} s), {cores} threads"},
            "one_thread": {"value": round(v_one, 2), "cores": 1, "sample": f"{n3} steps ({dt3:.1f} s), dropout {c['dropout']}"}}


def spawn_ranks(n):
    """``--gpus N`` run by hand (no WORLD_SIZE): start the N rank processes -- fresh children, before this parent has
    touched the GPU -- hand them the rendezvous through the environment torch.distributed.run would set, relay rank 0's
    JSON line (the children inherit stdout) and exit with the worst return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", type=int, default=3, choices=[1, 3, 8],
                    help="3: split-bf16 (parity grade, default); 1: single bf16 pass; 8: fp8 forward products (configs[4]'s \"fp8 MFMA weights\")")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grid", action="store_true", help="skip the folds/hr leg")
    ap.add_argument("--fits-per-gpu", type=int, default=5,
                    help="host threads per GPU in the grid leg (each runs work units); measured round 4 on one MI355X, same sample: "
                         "4 / 5 / 6 / 8 threads -> 30.7 / 31.3 / 31.0 / 30.7 k folds/hr (the GPU is the limit from 4)")
    ap.add_argument("--lockstep", type=int, default=15, help="fits per work unit, advanced through one launch sequence, in the grid leg")
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                    help="hipGraph replay, plain stream launches, or time both during warmup and keep the faster (default)")
    ap.add_argument("--eager", action="store_true", help="same as --launch eager")
    ap.add_argument("--dropout", type=float, default=None, help="override the workload's dropout rate (0: what the dropout epilogues cost)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))      # the parent has not touched the GPU (device_count() does not initialise it)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only compute path")
    ndev = torch.cuda.device_count()
    shared = world > ndev                      # rehearsal on a box with fewer GPUs than ranks: ranks share GPUs, gloo barrier
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)           # before the process group: RCCL binds its communicator to the current device
    backend = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", timeout=datetime.timedelta(hours=2))    # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=datetime.timedelta(hours=2))
        backend = dist.get_backend()
    dev = torch.device("cuda", dev_index)
    rccl_ranks = 0
    if backend == "nccl":
        # count the ranks THROUGH RCCL (an all-reduce of ones on the GPUs), do not assume them: the bench line's ranks.rccl_ranks is
        # this sum, and a communicator that does not span the world stops the run here instead of producing a line
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: RCCL all-reduce counted {rccl_ranks} ranks, WORLD_SIZE is {world}")

    from slnlp import synth, tf_engine as te
    c = dict(WORKLOADS[args.workload], precision=args.precision)
    if args.dropout is not None:
        c["dropout"] = args.dropout
    B, S = c["B"], c["S"]
    cfg, sd = build_sd(c, seed=1 + rank)
    if "rnn" in c:
        from slnlp import rnn_engine as re_
        eng = re_.RnnEngine(cfg, device=dev, seed=1 + rank)
    else:
        eng = te.TransformerEngine(cfg, device=dev, seed=1 + rank)
    eng.load_state(sd)
    eng.set_lr(LR)
    # synthetic dataset, resident in HBM before the timed region (SURVEY.md section 8d recipe)
    n_batches = 200
    Xn, Ln, yn = synth.make_batch(n_batches * B, S, c["Vs"], c["Vt"], seed=1 + rank)
    Xd, yd, Ld = torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev), torch.from_numpy(Ln).to(dev)

    stream = torch.cuda.Stream(device=dev)

    launch = {"auto": "auto", "graph": True, "eager": False}["eager" if args.eager else args.launch]
    from slnlp.launch import PROBE
    # "auto" times a few steps of hipGraph replay and of plain launches and keeps the faster (slnlp/launch.py: 2 + 2 * PROBE steps).
    # With a short --warmup (the driver's 5) those steps run as a fixed PRE-warmup in front of it -- untimed like the warmup itself,
    # so the W warmup steps and the K timed steps all run in the chosen mode
    pre_warmup = max(0, 2 + 2 * PROBE - args.warmup) if launch == "auto" else 0

    def run(k0, k):
        for i in range(k0, k0 + k):
            j = (i % n_batches) * B
            eng.step(Xd[j:j + B], yd[j:j + B], Ld[j:j + B], MOMENTUM, MAX_NORM, graph=launch)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        if pre_warmup:
            run(n_batches - pre_warmup, pre_warmup)
        run(0, args.warmup)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(args.warmup, args.steps)
        ev1.record(stream)
        barrier()
        wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    loss_end = eng.loss
    rank_walls = [wall]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device="cpu" if shared else dev)
        t[rank] = wall
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_walls = [float(v) for v in t.cpu()]
        wall = max(rank_walls)

    # outside the timed region: the dominant kernel's launches timed inside eager train steps (rank 0; roofline.population)
    in_step = None
    if rank == 0 and "rnn" not in c:
        def run_eager(k):
            with torch.cuda.stream(stream):
                for i in range(k):
                    j = (i % n_batches) * B
                    eng.step(Xd[j:j + B], yd[j:j + B], Ld[j:j + B], MOMENTUM, MAX_NORM, graph=False)
        try:
            in_step = in_step_plane_launches(run_eager)
        except Exception as e:                    # the measurement hook must never take the bench line down
            print(f"[bench] in-step launch timing failed: {e}", file=sys.stderr)

    # ---- leg 2: the sharded grid search, every rank takes part (collectives inside)
    chosen = eng._launch.mode((B, float(MOMENTUM), float(MAX_NORM))) if launch == "auto" else None
    grid = None
    if "rnn" not in c and not args.no_grid:
        del eng
        torch.cuda.empty_cache()
        grid = grid_folds_per_hour(dev, world, rank, fits_per_gpu=args.fits_per_gpu, lockstep=args.lockstep)

    out = None
    if rank == 0:
        seqs = world * B * args.steps
        if launch == "auto":
            launch_used = (f"{'hipGraph replay' if chosen == 'graph' else 'eager stream launches'} (auto: timed both in "
                           f"{'warmup' if not pre_warmup else str(pre_warmup) + ' untimed pre-warmup steps + warmup'})")
        else:
            launch_used = "hipGraph replay" if launch else "eager stream launches"
        step_flops = 3.0 * fwd_flops_per_seq(c) * B
        ms_event = ev_ms / args.steps           # HIP events on the launch stream, rank 0
        achieved = step_flops / (ms_event * 1e-3) / 1e12
        # parity on a held-out batch (eval mode) against the CPU oracle
        cfg0, sd0 = build_sd(c, seed=1)
        Xe, ye, Le = torch.from_numpy(Xn[:B]), torch.from_numpy(yn[:B]), torch.from_numpy(Ln[:B])
        if "rnn" in c:
            e2 = re_.RnnEngine(cfg0, device=dev)
            e2.load_state(sd0)
            lp = e2.forward(Xe.to(dev), ye.to(dev), Le.to(dev)).cpu()
        else:
            e2 = te.TransformerEngine(cfg0, device=dev)
            e2.load_state(sd0)
            lp = e2.forward(Xe.to(dev), ye.to(dev)).cpu()
        parity = cpu_parity(c, sd0, Xe, ye, Le, lp)
        out = {
            "metric": f"train seq/s (batch={B},len={S})", "value": round(seqs / wall, 1), "unit": "seq/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {1: "bf16", 3: "bf16x3", 8: "fp8 (forward products) + bf16x3 (backward)"}[args.precision],
            "data": "synthetic (numpy seed recipe: ids, lengths, labels; seed-recipe weights)",
            "config": {"workload": (f"{args.workload}: EncoderDecoder{c['rnn'].upper()}Attn train step E{c['E']} Hd{c['Hd']} N{c['N']} "
                                    if "rnn" in c else
                                    f"{args.workload}: Transformer train step E{c['E']} H{c['H']} N{c['N']} F{c['F']} ") +
                                   f"batch {B} len {S} |src| {c['Vs']} |tgt| {c['Vt']} dropout {c['dropout']}, "
                                   "fwd+CE+bwd+clip(0.5)+SGD(m=.9)",
                       "launch": launch_used, "per_gpu": "independent fit (grid shard)" + (f"; REHEARSAL: {world} ranks share {ndev} GPU(s)" if shared else "")},
            "ranks": {"world": world, "backend": backend, "rccl_ranks": rccl_ranks,
                      "devices_visible": ndev, "rank_wall_s": [round(v, 4) for v in rank_walls]},
            "roofline_step": {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_DENSE_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(achieved / BF16_DENSE_PEAK_TFLOPS, 5),
                              "traffic": pmc_traffic(args.workload)[0], "traffic_source": pmc_traffic(args.workload)[1],
                              "launch": "one train step (all its kernels)", "flops_per_launch": step_flops,
                              "ms_per_launch_hip_events": round(ms_event, 4)},
            "parity": parity, "final_loss": round(loss_end, 5),
        }
        if grid is not None:
            out["grid"] = grid
        if "rnn" not in c:
            with torch.cuda.stream(stream):
                dk = dominant_kernel_roofline(c, 3 if args.precision == 8 else args.precision, dev, args.workload, in_step)
                if args.precision == 8:
                    out["roofline_fp8"] = fp8_kernel_roofline(c, dev)
            if dk:
                out["roofline"] = dk
                with torch.cuda.stream(stream):
                    out["roofline_large_launch"] = large_launch_roofline(3 if args.precision == 8 else args.precision, dev)
        if world == 1 and not args.no_cpu_baseline and args.precision != 8:
            out["concurrent_fits"] = concurrent_fits(c, args.precision, dev)
        out.setdefault("roofline", dict(out["roofline_step"]))   # RNN workloads: no single dominant GEMM, the step is the unit
        if not args.no_cpu_baseline and world == 1:       # reported at N = 1 only (the other ranks would just wait)
            out["cpu_baseline"] = cpu_baseline(c, sd0, torch.from_numpy(Xn), torch.from_numpy(yn), torch.from_numpy(Ln))
            out["gpu_over_cpu"] = round(out["value"] / world / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
