"""Benchmark of the coupling-flow hot path on MI355X (contract: see the round brief).

Metric (BASELINE.json): samples/s + mean-log-prob error vs CPU on configs[1]:
9 x AffineHalfFlow ("RNVP"), d=64, batch 2^20 synthetic Gaussian, fp32.

A step = one density evaluation of one batch resident in HBM, through the drop-in API:
    zs, log_det = NormalizingFlow.inverse(x)        (9 fused coupling kernels, log_det += ld)
    log p = log_det + N(0,I).log_prob(zs[-1])       (epilogue kernel, fp64 sum over rows)
followed, when N > 1, by one RCCL all-reduce of (sum, count).  Every rank holds its own 2^20
rows (weak scaling); value = rows of all ranks / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c3|c2t|c3t|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)

WORKLOADS = {
    # name: (dim, rows per GPU, description)
    "c2": (64, 1 << 20, "9xAffineHalfFlow d=64 batch=2^20 inverse+log_prob (BASELINE configs[1])"),
    "c4": (256, 1 << 19, "9xAffineHalfFlow d=256 batch=2^19/GPU inverse+log_prob (BASELINE configs[3]: 2^22 rows over "
                         "8 GPUs = this shard per GPU; at --gpus 8 the line IS configs[3])"),
    "c3": (32, 1 << 20, "3x[ActNorm,Glow,NSF_CL] d=32 K=8 n_h=8 batch=2^20 inverse+log_prob (BASELINE configs[2])"),
    "c6": (512, 1 << 18, "9xAffineHalfFlow d=512 batch=2^18 inverse+log_prob: a shape without per-shape kernel (they end "
                         "at d = 256), one launch per layer on the run-time-shaped matrix-core kernel (round 6: the "
                         "any-shape path; round 5 ran this on the VALU kernel at 18 ns per row and layer)"),
    "c2f": (64, 1 << 20, "FusedAffineStack(9xAffineHalfFlow) d=64 batch=2^20 inverse+log_prob (opt-in whole-stack "
                         "fusion: no intermediates; reported separately from c2)"),
    "c3f": (32, 1 << 20, "3xFusedSplineBlock[ActNorm,Glow,NSF_CL] d=32 K=8 n_h=8 batch=2^20 inverse+log_prob "
                         "(opt-in fusion: block intermediates not materialised; reported separately from c3)"),
    "c1": (2, 4096, "9xAffineHalfFlow d=2 batch=4096 (half-moons) inverse+log_prob: LATENCY per pass, eager and replayed "
                    "from a hipGraph (BASELINE configs[0]: SURVEY 8d 'report latency, not roofline')"),
    "c5": (800, 512 * 500, "MNFLinear(800,50).sample_z: 2xRNVP d=800 h=50 on 512x500 MC rows (BASELINE configs[4])"),
    "c5b": (50, 512 * 500, "MNFLinear(50,10).sample_z: 2xRNVP d=50 h=50 on 512x500 MC rows (the second MNFLinear of "
                           "BASELINE configs[4]'s MNF-LeNet)"),
    "c2t": (64, 1 << 20, "training step of 9xAffineHalfFlow d=64 batch=2^20: -mean log-prob, backward, Adam (SURVEY 8f "
                         "rank 1; the reference's tests train through these layers, tests/test_flows.py:14-31)"),
    "c5t": (800, 512 * 500, "training step of MNFLinear(800,50) on 512x500 MC rows: sample_z (2xRNVP d=800 h=50), forward, "
                            "mean-square loss, backward, Adam (BASELINE configs[4]'s caller under loss.backward(): what "
                            "tests/test_mnf_mnist.py:14-56 trains through)"),
    "lenet": (784, 128, "training step of MNF-LeNet (MNFConv2d(1,20,5), MNFConv2d(20,50,5), MNFLinear(800,50), "
                        "MNFLinear(50,10)) on a batch of 128 28x28 images: kl_div of the four layers + NLL, backward, Adam "
                        "(what tests/test_mnf_mnist.py:14-56 does per batch), replayed from one hipGraph"),
    "c3t": (32, 1 << 20, "training step of 3x[ActNorm,Glow,NSF_CL] d=32 K=8 n_h=8 batch=2^20: -mean log-prob, backward, "
                         "Adam (SURVEY 8f rank 1 for BASELINE configs[2]'s model; tests/test_flows.py:89-99 trains it)"),
}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table


def build_c3(device):
    """3 x [ActNorm, Glow, NSF_CL(32, K=8, B=3, n_h=8)] with fixture-recipe parameters."""
    from torch_mnf_amd import synthetic as recipes
    import torch_mnf_amd as amd

    flows, layers = [], []
    for i in range(3):
        an = amd.ActNormFlow(32)
        ap = recipes.actnorm_params(630 + i, 32)
        an.load_state_dict(ap)
        an.data_dep_init_done = True
        gl = amd.Glow(32)
        gp = recipes.glow_params(600 + i, 32)
        gl.P = gp["P"]
        gl.load_state_dict({"L": gp["L"], "S": gp["S"], "U": gp["U"]})
        sp = amd.NSF_CL(32, K=8, B=3, n_h=8)
        sd = recipes.nsf_cl_params(610 + i, 32, 8, 8)
        sp.load_state_dict(sd)
        flows += [an, gl, sp]
        layers += [{"kind": "affine_const", "params": ap}, {"kind": "glow", "params": gp},
                   {"kind": "nsf_cl", "K": 8, "B": 3.0, "params": sd}]
    model = amd.NormalizingFlowModel(amd.StandardNormal(32, device), flows).to(device)
    return model, layers


def build_model(dim: int, device):
    from torch_mnf_amd import synthetic as recipes
    import torch_mnf_amd as amd

    flows = []
    sds = recipes.c2_stack_params(dim)
    for i, sd in enumerate(sds):
        f = amd.AffineHalfFlow(dim, parity=bool(i % 2))
        f.load_state_dict(sd)
        flows.append(f)
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim, device), flows).to(device)
    layers = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd} for i, sd in enumerate(sds)]
    return model, layers


def pmc_traffic(workload: str) -> tuple[float | None, str | None]:
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  The counters cannot be
    collected inside a timed run (rocprofv3 PMC passes serialise the kernels), so this is READ from the committed
    PMC profile of this same command -- profiles/r<N>/<workload>_pmc_traffic.json, newest round first: FETCH_SIZE
    doubled per the gfx950 rule, plus WRITE_SIZE -- and `traffic_source` in the JSON line says so.  (None, None)
    when no such profile is committed."""
    rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()),
                    key=lambda d: -int(d[1:])) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    for r in rounds:
        path = os.path.join(ROOT, "profiles", r, f"{workload}_pmc_traffic.json")
        try:
            with open(path) as fh:
                return json.load(fh)["traffic_bytes_per_launch"], f"committed rocprofv3 --pmc profile profiles/{r}/{workload}_pmc_traffic.json (not measured in this run)"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def valu_issue(workload: str) -> dict | None:
    """The committed SQ-counter digest of the dominant kernel (tools/make_valu_json.py): share of the chip's vector
    issue capacity it used.  Read, not measured in this run (PMC passes serialise the kernels)."""
    rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()),
                    key=lambda d: -int(d[1:])) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    for r in rounds:
        try:
            with open(os.path.join(ROOT, "profiles", r, f"{workload}_valu_issue.json")) as fh:
                d = json.load(fh)
            d["file"] = f"profiles/{r}/{workload}_valu_issue.json"
            return d
        except (OSError, ValueError):
            continue
    return None


def apply_valu_convention(r: dict, workload: str, avg_kernel_s: float) -> None:
    """Re-express a roofline object whose kernel is bound by vector-instruction issue (the spline kernels, SURVEY 8(d):
    264 B per row against thousands of vector instructions per 16 rows): `bound` "valu", achieved / peak / frac in wave
    instructions per second, the HBM figures kept under their own names, and next to them what the counters MEASURED --
    how long the SIMDs' vector units were occupied (`simd_valu_busy_frac`) and the matrix pipe (`mfma_pipe_frac`).
    Instruction counts come from the committed SQ-counter digest (tools/make_valu_json.py), time from THIS run's HIP
    events.  One convention for c3 and c3t."""
    r.update({"hbm_achieved_GBps": r["achieved"], "hbm_frac": r["frac"], "hbm_peak_GBps": r["peak"]})
    vi = valu_issue(workload)
    if vi is None:
        r.update({"bound": "valu", "valu_source": None,
                  "note": "no committed *_valu_issue.json: achieved / frac are the HBM figures"})
        return
    cycles_now = vi["effective_clock_GHz"] * 1e9 * avg_kernel_s
    plain = vi["SQ_INSTS_VALU"] - vi["SQ_INSTS_MFMA"]
    ginstr = vi["SQ_INSTS_VALU"] / avg_kernel_s / 1e9
    peak = 1024 / 2.0 * vi["effective_clock_GHz"]
    r.update({"bound": "valu", "achieved": ginstr, "peak": peak, "unit": "G wave-instr/s",
              "frac": (2.0 * plain + 8.0 * vi["SQ_INSTS_MFMA"]) / (1024 * cycles_now),
              "frac_plain_count": ginstr / peak,
              "simd_valu_busy_frac": vi.get("simd_valu_busy_frac"), "mfma_pipe_frac": vi.get("mfma_pipe_frac"),
              "valu_source": f"{vi['file']} (SQ_INSTS_VALU {vi['SQ_INSTS_VALU']:.4g}, SQ_INSTS_MFMA "
                             f"{vi['SQ_INSTS_MFMA']:.4g} per launch, effective clock "
                             f"{vi['effective_clock_GHz']:.2f} GHz from GRBM_GUI_ACTIVE; committed "
                             "rocprofv3 --pmc profile, not measured in this run)",
              "valu_definition": "frac = (2 cyc x plain vector instr + 8 cyc x MFMA) / (1024 SIMDs x kernel "
                                 "cycles): share of SIMD issue cycles used; peak = one wave64 vector "
                                 "instruction per SIMD per 2 cycles; simd_valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / "
                                 "the same SIMD cycles (how long the vector units were occupied, profiled run)"})


def apply_transcendental_floor(r: dict, workload: str, rows: int, ms_per_step: float) -> None:
    """The floor no instruction diet of the plain arithmetic can go below: the transcendental wave-instructions of a step
    (v_exp / v_log / v_rcp / v_sqrt per spline element, counted from the kernels' device assembly by
    tools/transcendental_count.py and committed as profiles/rN/<workload>_transcendentals.json) at the guide's issue cost
    of 8 cycles each on 1,024 SIMDs at the 2.4 GHz peak clock (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')."""
    rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()),
                    key=lambda d: -int(d[1:])) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    for rd in rounds:
        path = os.path.join(ROOT, "profiles", rd, f"{workload}_transcendentals.json")
        try:
            with open(path) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        wave_instr = t["per_element_per_step_layer"] * t["elements_per_row"] * t["layers_per_step"] * rows / 64.0
        floor_ms = wave_instr * t["cycles_per_transcendental"] / 1024.0 / 2.4e9 * 1e3
        r.update({"transcendental_floor_ms": floor_ms, "frac_of_transcendental_floor": floor_ms / ms_per_step,
                  "transcendentals_per_element": t["per_element_per_step_layer"],
                  "plain_valu_per_element": t.get("valu_per_element"),
                  "transcendental_source": f"profiles/{rd}/{workload}_transcendentals.json ({t.get('what', '')}); floor = count x "
                                           "elements x rows / 64 lanes x 8 cycles / 1,024 SIMDs / 2.4 GHz, against ms_per_step"})
        return


def physical(traffic, avg_kernel_s: float) -> dict:
    """The HBM fraction by bytes that actually cross the interface (PMC), next to the SURVEY 8(d) algorithmic one."""
    if not traffic:
        return {"achieved_physical": None, "frac_physical": None}
    gbs = traffic / avg_kernel_s / 1e9
    return {"achieved_physical": gbs, "frac_physical": gbs / HBM_PEAK_GBS}


def cpu_baseline(layers, dim: int, x_sample_gpu: torch.Tensor, budget_s: float = 12.0) -> tuple[dict, float, int]:
    """Time the CPU oracle (the restated reference path: stock PyTorch CPU ops, all host
    threads) on a bounded sample of the same workload.  Checker and baseline only."""
    from oracle import flow_oracle as O

    # Pick the ATen thread count the way a user of the reference would tune it on this host:
    # the op sequence is many small ops, so "all cores" is rarely the fastest setting.
    host = os.cpu_count() or 1
    probe = 1 << 14
    xp = x_sample_gpu[:probe].cpu()
    best_rate, cores = 0.0, 1
    with torch.no_grad():
        for n in sorted({c for c in (8, 16, 32, 64, host) if c <= host}):
            torch.set_num_threads(n)
            O.mean_log_prob(xp[:2048], layers)  # warm the pool
            t0 = time.perf_counter()
            O.mean_log_prob(xp, layers)
            r = probe / (time.perf_counter() - t0)
            if r > best_rate:
                best_rate, cores = r, n
            if probe / r > 3.0:  # this setting is already hopeless; larger pools only get worse
                break
    torch.set_num_threads(cores)
    rows = int(min(x_sample_gpu.shape[0], max(probe, 1 << (int(best_rate * budget_s).bit_length() - 1))))
    x = x_sample_gpu[:rows].cpu()
    best = float("inf")
    mean = None
    with torch.no_grad():
        t_start = time.perf_counter()
        for _ in range(3):
            t0 = time.perf_counter()
            mean, _ = O.mean_log_prob(x, layers)
            best = min(best, time.perf_counter() - t0)
            if time.perf_counter() - t_start > 2.0 * budget_s:
                break
    # ... and on ONE thread on the SAME rows as the multi-thread figure when that takes under ~20 s, else
    # both on the first 65,536 rows (the two figures are then measured on one and the same sample either way)
    torch.set_num_threads(1)
    with torch.no_grad():
        O.mean_log_prob(x[:1024], layers)
        t0 = time.perf_counter()
        O.mean_log_prob(x[:8192], layers)
        est = (time.perf_counter() - t0) / 8192 * rows
    n1 = rows if est < 20.0 else int(min(rows, 1 << 16))
    with torch.no_grad():
        t0 = time.perf_counter()
        O.mean_log_prob(x[:n1], layers)
        one_thread = n1 / (time.perf_counter() - t0)
    torch.set_num_threads(cores)
    if n1 == rows:
        multi_same = rows / best
    else:
        with torch.no_grad():
            O.mean_log_prob(x[:n1], layers)
            t0 = time.perf_counter()
            O.mean_log_prob(x[:n1], layers)
            multi_same = n1 / (time.perf_counter() - t0)
    info = {
        "value": rows / best,
        "unit": "samples/s",
        "cores": cores,
        "torch_num_threads": torch.get_num_threads(),
        "one_thread": {"value": one_thread, "unit": "samples/s", "rows": n1,
                       "multi_thread_on_the_same_rows": multi_same,
                       "sample": f"first {n1} rows (the SAME rows as `multi_thread_on_the_same_rows`, {cores} ATen "
                                 f"threads), one ATen thread, one pass"},
        "host_cores": host,
        "kind": "port",
        "sample": f"oracle (PyTorch-CPU restatement of the reference path), first {rows} rows of the "
                  f"rank-0 batch, one inverse pass + base log-prob, best of <=3, {best:.3f} s, "
                  f"{cores} ATen threads (fastest of 8/16/32/64/{host} on a probe)",
    }
    return info, mean, rows


# the coupling kernels evaluate fp32 products on the f16 matrix pipe (hi + lo split, three MFMAs per product,
# fp32 accumulation; csrc/mnf_split.h); MNF_FP32_MFMA=1 switches to the fp32 MFMA kernels for A/B runs
SPLIT = os.environ.get("MNF_FP32_MFMA", "0") != "1"
ARITHMETIC = ("fp32 via split f16 MFMA (hi+lo, 3 products per fp32 product, fp32 accumulate; fp32-MFMA fallback for "
              "operands outside the f16 range)") if SPLIT else "fp32 MFMA"
AHF_KERNEL = "ahf_split_kernel" if SPLIT else "ahf_mfma_kernel"


def main_c5(args, rank, world, device, dim, rows, desc, n_out: int = 50) -> None:
    """Config 5: the flow_q of MNFLinear(800, 50) -- c5b: of MNFLinear(50, 10) -- on 256,000 MC rows.  A step = one
    sample_z call (prologue kernel + two seeded RNVP kernels, log-det accumulated in-kernel); fp32-MFMA bound."""
    from torch_mnf_amd import synthetic as recipes
    import torch_mnf_amd as amd

    layer = amd.MNFLinear(dim, n_out)
    for i, f in enumerate(layer.flow_q.flows):
        f.load_state_dict(recipes.rnvp_params(800 + i, dim, 50))
    layer.to(device)
    gen = torch.Generator(device=device).manual_seed(4321 + rank)
    eps = torch.randn(rows, dim, device=device, generator=gen)  # noise resident in HBM; masks are generated in-kernel

    def step():
        return layer.sample_z(rows, eps=eps)

    with torch.no_grad():
        step()
        torch.cuda.synchronize()
        t_prime = time.perf_counter()
        while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
            step()
            torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        layer.flow_q.layer_events = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gc.enable()
        events, layer.flow_q.layer_events = layer.flow_q.layer_events, None
    backend = os.environ.get("MNF_BENCH_BACKEND", "nccl")
    elapsed, per_rank_s = rank_times(elapsed, world, device, backend)
    if rank == 0:
        kern_ms = [a.elapsed_time(b) for a, b, *_ in events]
        avg_s = sum(kern_ms) / len(kern_ms) / 1e3
        flops = 2 * (dim * 50 + 2 * 50 * dim) * rows  # 240,000 per row
        tf = flops / avg_s / 1e12
        out = {
            "metric": f"rows/s, {desc}", "value": world * rows * args.steps / elapsed, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "rows_per_gpu": rows, "dim": dim, "hidden": [50], "mask": "in-kernel (seeded)",
                       "primed_ms": args.prime_ms, "arithmetic": ARITHMETIC, "total_rows": world * rows},
            "distributed": dist_info(world, backend, per_rank_s, args.steps),
        }
        # Algorithmic bytes per row with the in-kernel mask: read z (4d), write x (4d), log_det RMW (8).  The
        # register-resident kernel moves exactly that; the streaming kernel (MNF_RNVP_RESIDENT=0) reads z a second
        # time for the gate epilogue: 12d + 8.
        algo_bytes = (8 * dim + 8) * rows
        # rows held in registers, z read once: rnvp_resident_kernel at d = 800, rnvp_narrow_kernel at d <= 64 (c5b)
        resident = SPLIT and os.environ.get("MNF_RNVP_RESIDENT", "1") != "0" and dim == 800
        narrow = SPLIT and dim <= 64
        if SPLIT:  # memory-path bound (tools ablations: no MFMAs -> same time), priced against HBM
            gbs = algo_bytes / avg_s / 1e9
            traffic, source = pmc_traffic("c5" if resident else "c5b" if dim != 800 else "c5_streaming")
            out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
                               **physical(traffic, avg_s),
                               "kernel": ("rnvp_resident_kernel<50,50> (rows resident in the register file, z read once)"
                                          if resident else
                                          "rnvp_narrow_kernel<50,seeded> (16 waves per CU, operand image resident in LDS, "
                                          "a row tile in registers, z read once)" if narrow else
                                          "rnvp_split_kernel<50,seeded> (z read twice)"),
                               "avg_kernel_us": avg_s * 1e6,
                               "algorithmic_bytes_per_launch": algo_bytes, "launches_timed": len(kern_ms),
                               "bytes_moved_per_launch_by_design": (8 if resident or narrow else 12) * dim * rows + 8 * rows,
                               "fp32_equivalent_tflops": tf}
        else:
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                               "kernel": "rnvp_mfma_kernel<50,seeded>", "avg_kernel_us": avg_s * 1e6,
                               "algorithmic_flops_per_launch": flops, "launches_timed": len(kern_ms),
                               "algorithmic_GBps": (12 * dim + 8) * rows / avg_s / 1e9}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import flow_oracle as O

            n = 1 << 17
            torch.set_num_threads(min(32, os.cpu_count() or 1))
            q0m, q0v = layer.q0_mean.detach().cpu(), layer.q0_log_var.detach().cpu()
            e_cpu = eps[:n].cpu()
            seeds = (11, 12)
            masks = [layer.flow_q.flows[i].mask_for(seeds[i], n, device).cpu() for i in range(2)]
            specs = [{"kind": "rnvp", "params": recipes.rnvp_params(800 + i, dim, 50), "mask": masks[i]} for i in range(2)]
            best = float("inf")
            with torch.no_grad():
                for _ in range(3):
                    t1 = time.perf_counter()
                    z_cpu, ld_cpu = O.sample_z(q0m, q0v, e_cpu, specs)
                    best = min(best, time.perf_counter() - t1)
                z_gpu, ld_gpu = layer.sample_z(n, eps=eps[:n].contiguous(), masks=[m.to(device) for m in masks])
            err = float((z_gpu.cpu() - z_cpu).abs().max() / z_cpu.abs().max())
            out["cpu_baseline"] = {"value": n / best, "unit": "rows/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"oracle sample_z on the first {n} rows (masks materialised from the "
                                             f"library's generator), best of 3, {best:.3f} s"}
            out["parity"] = {"rows": n, "z_normwise_err": err,
                             "log_det_normwise_err": float((ld_gpu.cpu() - ld_cpu).abs().max() / ld_cpu.abs().max()),
                             "tolerance": 1e-5}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main_c1(args, rank, world, device, dim, rows, desc) -> None:
    """Config 1 (the reference's own CPU-runnable case, examples/half_moons.ipynb): nine AffineHalfFlow layers on 4,096
    two-dimensional points.  At this size a pass is nine launches of a few microseconds: what is reported is the LATENCY
    of one inverse + log-prob pass, eager and replayed from a hipGraph, not a roofline fraction."""
    if world != 1:
        raise SystemExit("--workload c1 is a single-GPU latency measurement")
    import torch_mnf_amd as amd
    from torch_mnf_amd import synthetic as recipes

    torch.manual_seed(11)
    flows = []
    for i, sd in enumerate(recipes.c2_stack_params(dim, 9)):
        f = amd.AffineHalfFlow(dim, bool(i % 2))
        f.load_state_dict(sd)
        flows.append(f)
    model = amd.NormalizingFlowModel(amd.StandardNormal(dim), flows).to(device)
    gen = torch.Generator(device=device).manual_seed(99)
    x = torch.randn(rows, dim, device=device, generator=gen)

    def timed(fn, n):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        t_prime = time.perf_counter()
        while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
            model.log_prob(x, return_sum=True)
            torch.cuda.synchronize()
        eager_s = timed(lambda: model.log_prob(x, return_sum=True), args.steps)
        _, tot = model.log_prob(x, return_sum=True)
        replay = model.graphed_log_prob(x)
        graph_s = timed(lambda: replay(x), args.steps)
        _, tot_g = replay(x)
    out = {
        "metric": f"samples/s, {desc}", "value": rows / eager_s, "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": eager_s * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": rows, "dim": dim, "n_layers": 9, "total_rows": rows},
        "latency_us": {"eager_pass": eager_s * 1e6, "graphed_pass": graph_s * 1e6,
                       "graphed_samples_per_s": rows / graph_s,
                       "note": "one inverse + log-prob pass over 4,096 rows: nine launches + the epilogue; launch-bound"},
        "roofline": {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                     "kernel": "(launch-bound: latency reported instead, SURVEY 8d)", "avg_kernel_us": None},
        "mean_log_prob": float(tot) / rows, "mean_log_prob_graphed": float(tot_g) / rows,
    }
    if not args.no_cpu_baseline:
        from oracle import flow_oracle as O

        layers = [{"kind": "affine_half", "parity": bool(i % 2), "params": sd}
                  for i, sd in enumerate(recipes.c2_stack_params(dim, 9))]
        torch.set_num_threads(1)
        xc = x.cpu()
        with torch.no_grad():
            O.mean_log_prob(xc, layers)
            t0 = time.perf_counter()
            for _ in range(20):
                mean, _ = O.mean_log_prob(xc, layers)
            cpu_s = (time.perf_counter() - t0) / 20
        out["cpu_baseline"] = {"value": rows / cpu_s, "unit": "samples/s", "cores": 1, "kind": "port",
                               "latency_us": cpu_s * 1e6,
                               "sample": "oracle, the same 4,096 rows, one ATen thread, mean of 20 passes"}
        out["parity"] = {"mean_log_prob_rel_err": abs(float(tot) / rows - float(mean)) / abs(float(mean)), "tolerance": 1e-5}
    print(json.dumps(out))


def main_train(args, rank, world, device, dim, rows, desc) -> None:
    """SURVEY 8f rank 1 on the bench contract: a step = one Adam step of the C2 model on the resident batch -- forward
    through the stack kernel (every intermediate kept for the backward pass), -mean log-prob, backward layer by layer
    (mnf_affine_half_bwd_split + its fix-up pass), one fused Adam launch over the flat parameter buffer.  The dominant
    kernel is the gradient kernel: x and grad_y in, grad_x out per layer = (12 d + 4) bytes per row per launch."""
    if world != 1:
        raise SystemExit("--workload c2t measures one GPU (data-parallel training would add a gradient all-reduce)")
    import torch_mnf_amd as amd
    from torch_mnf_amd import flows as amd_flows

    model, layers = build_model(dim, device)
    opt = amd.FusedAdam(amd.FlatParameters(model), lr=1e-4)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(rows, dim, device=device, generator=gen)  # resident in HBM before timing
    loss_box = [None]

    def step():
        opt.zero_grad()
        loss = -model.log_prob(x).mean()
        loss.backward()
        opt.step()
        loss_box[0] = loss

    step()
    torch.cuda.synchronize()
    first_loss = float(loss_box[0])
    t_prime = time.perf_counter()
    while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
        step()
        torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    amd_flows.bwd_kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    events, amd_flows.bwd_kernel_events = amd_flows.bwd_kernel_events, None
    kern_ms = [a.elapsed_time(b) for a, b in events]
    split_kernel = len(kern_ms) > 0
    out = {
        "metric": f"samples/s, {desc}", "value": rows * args.steps / elapsed, "unit": "samples/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": rows, "dim": dim, "layers": len(model.flows), "hidden": [24, 24, 24],
                   "optimizer": "FusedAdam over FlatParameters (one buffer, one launch)", "primed_ms": args.prime_ms,
                   "arithmetic": ARITHMETIC, "total_rows": rows},
        "distributed": dist_info(1, "nccl", [elapsed], args.steps),
        "loss_first_step": first_loss, "loss_last_step": float(loss_box[0]),
    }
    if split_kernel:
        avg_s = sum(kern_ms) / len(kern_ms) / 1e3
        algo_bytes = (12 * dim + 4) * rows
        traffic, source = pmc_traffic("c2t")
        gbs = algo_bytes / avg_s / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": source, **physical(traffic, avg_s),
                           "kernel": "ahf_bwd_split_kernel<32,24,inverse> (one layer's gradients per launch: recompute, "
                                     "delta chain, weight gradients; 9 launches per step)",
                           "avg_kernel_us": avg_s * 1e6, "algorithmic_bytes_per_launch": algo_bytes,
                           "launches_timed": len(kern_ms), "launches_per_step": len(model.flows),
                           "note": "issue / latency bound at one wave per SIMD (DESIGN.md 3.5); the HBM figures say how "
                                   "far the kernel is from the traffic it has to move"}
    else:
        out["roofline"] = {"bound": "mfma", "achieved": None, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                           "traffic": None, "kernel": "ahf_bwd_mfma_kernel<32,24> (MNF_BWD_FP32=1 / MNF_FP32_MFMA=1)"}
    if not args.no_cpu_baseline:
        # the oracle's training step on the host: forward through the restated reference path, torch.autograd backward
        # (the parameter update is negligible next to it), on a bounded slice of the same batch
        from oracle import flow_oracle as O

        n = 1 << 16
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        xs = x[:n].cpu()
        specs = [{"kind": "affine_half", "parity": l["parity"],
                  "params": {k: v.clone().requires_grad_(True) for k, v in l["params"].items()}} for l in layers]
        best = float("inf")
        for _ in range(2):
            for sp in specs:
                for v in sp["params"].values():
                    v.grad = None
            t1 = time.perf_counter()
            zs, ld = O.flow_stack(xs, specs, inverse=True)
            loss_cpu = -(ld + O.std_normal_log_prob(zs[-1])).mean()
            loss_cpu.backward()
            best = min(best, time.perf_counter() - t1)
        out["cpu_baseline"] = {"value": n / best, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle forward + torch.autograd backward of -mean log-prob on the first {n} rows "
                                         f"(initial parameters), best of 2, {best:.3f} s"}
        # parity of the gradients: a fresh model with the same initial parameters on the same rows, against the
        # oracle evaluated in float64 (sums over 65,536 rows: the fp32 oracle's own rounding is ~1e-4 there)
        specs = [{"kind": "affine_half", "parity": l["parity"],
                  "params": {k: v.double().requires_grad_(True) for k, v in l["params"].items()}} for l in layers]
        zs, ld = O.flow_stack(xs.double(), specs, inverse=True)
        loss_cpu = -(ld + O.std_normal_log_prob(zs[-1])).mean()
        loss_cpu.backward()
        fresh, _ = build_model(dim, device)
        floor, amd_flows._dispatch.BWD_SPLIT_MIN_ROWS = amd_flows._dispatch.BWD_SPLIT_MIN_ROWS, 0
        try:
            loss_gpu = -fresh.log_prob(x[:n].contiguous()).mean()
            loss_gpu.backward()
        finally:
            amd_flows._dispatch.BWD_SPLIT_MIN_ROWS = floor
        worst = 0.0
        for sp, f in zip(specs, fresh.flows):
            for name, prm in f.named_parameters():
                ref = sp["params"][name].grad
                worst = max(worst, float((prm.grad.cpu().double() - ref).abs().max() / ref.abs().max()))
        out["parity"] = {"rows": n, "reference": "oracle in float64", "worst_parameter_gradient_normwise_err": worst,
                         "tolerance": 5e-5,  # (tests/test_hip_autograd.py: the 9-layer run's gradient bar)
                         "loss_gpu_vs_cpu_rel_err": abs(float(loss_gpu) - float(loss_cpu)) / abs(float(loss_cpu))}
    print(json.dumps(out))


def main_train_c5(args, rank, world, device, dim, rows, desc) -> None:
    """Config 5's caller on the training side of the bench contract: a step = one Adam step of MNFLinear(800, 50) on the
    resident batch -- sample_z (prologue + two RNVP launches with autograd), forward (mnf_mnf_linear_fwd_train), a
    mean-square loss, backward (mnf_mnf_linear_bwd; per RNVP layer mnf_rnvp_bwd_mfma = launches A, B-ts, B-n + fix-up),
    one fused Adam launch.  The dominant kernel is B-ts (grad_z and the t / s weight gradients of one RNVP layer): z and
    grad_x in, grad_z out = (12 d + 4) bytes per row per launch."""
    if world != 1:
        raise SystemExit("--workload c5t measures one GPU (data-parallel training would add a gradient all-reduce)")
    import torch_mnf_amd as amd
    from torch_mnf_amd import flows as amd_flows
    from torch_mnf_amd import synthetic as recipes

    n_out = 50

    def build():
        torch.manual_seed(55)
        layer = amd.MNFLinear(dim, n_out)
        for i, f in enumerate(layer.flow_q.flows):
            f.load_state_dict(recipes.rnvp_params(800 + i, dim, 50))
        return layer.to(device)

    layer = build()
    opt = amd.FusedAdam(amd.FlatParameters(layer), lr=1e-4)
    gen = torch.Generator(device=device).manual_seed(2468 + rank)
    x = torch.rand(rows, dim, device=device, generator=gen)  # (activations: non-negative, as after the reference's ReLU)
    loss_box = [None]

    def step():
        opt.zero_grad()
        loss = layer.forward(x).pow(2).mean()
        loss.backward()
        opt.step()
        loss_box[0] = loss

    step()
    torch.cuda.synchronize()
    first_loss = float(loss_box[0])
    t_prime = time.perf_counter()
    while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
        step()
        torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    amd_flows.rnvp_bwd_kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    events, amd_flows.rnvp_bwd_kernel_events = amd_flows.rnvp_bwd_kernel_events, None
    kern_ms = [a.elapsed_time(b) for a, b in events]
    out = {
        "metric": f"rows/s, {desc}", "value": rows * args.steps / elapsed, "unit": "rows/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": rows, "dim": dim, "n_out": n_out, "hidden": [50],
                   "mask": "in-kernel (seeded)", "optimizer": "FusedAdam over FlatParameters (one buffer, one launch)",
                   "primed_ms": args.prime_ms, "arithmetic": ARITHMETIC, "total_rows": rows},
        "distributed": dist_info(1, "nccl", [elapsed], args.steps),
        "loss_first_step": first_loss, "loss_last_step": float(loss_box[0]),
    }
    if kern_ms:
        avg_s = sum(kern_ms) / len(kern_ms) / 1e3
        algo_bytes = (12 * dim + 4) * rows
        traffic, source = pmc_traffic("c5t")
        gbs = algo_bytes / avg_s / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": source, **physical(traffic, avg_s),
                           "kernel": "rnvp_bwd_ts_shared_kernel<50,seeded> (launch B-ts of one RNVP layer's gradient pass: s, t, "
                                     "gate, g_k, grad_z and the row sums dWt, dWs; 2 launches per step)",
                           "avg_kernel_us": avg_s * 1e6, "algorithmic_bytes_per_launch": algo_bytes,
                           "launches_timed": len(kern_ms), "launches_per_step": 2,
                           "note": "z and grad_x in, grad_z out; launch A of the same pass reads z twice and grad_x once "
                                   "more (DESIGN.md 3.5)"}
    if not args.no_cpu_baseline:
        # the oracle's training step on the host (sample_z + MNFLinear.forward restated, torch.autograd backward) on a
        # bounded slice, with the masks and both noise draws materialised so that the GPU leg below sees the same ones
        from oracle import flow_oracle as O

        n = 1 << 14
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        fresh = build()
        g = torch.Generator().manual_seed(97)
        eps_z, eps_o = torch.randn(n, dim, generator=g), torch.randn(n, n_out, generator=g)
        masks = [fresh.flow_q.flows[i].mask_for(21 + i, n, device).cpu() for i in range(2)]
        names = ["W_mean", "W_log_var", "b_mean", "b_log_var", "q0_mean", "q0_log_var"]

        def oracle_step(dt):
            p = {k: getattr(fresh, k).detach().cpu().to(dt).requires_grad_(True) for k in names}
            specs = [{"kind": "rnvp", "mask": masks[i].to(dt),
                      "params": {k: v.detach().cpu().to(dt).requires_grad_(True)
                                 for k, v in fresh.flow_q.flows[i].state_dict().items()}} for i in range(2)]
            z, _ = O.sample_z(p["q0_mean"], p["q0_log_var"], eps_z.to(dt), specs)
            o = O.mnf_linear_forward(x[:n].cpu().to(dt), z, p["W_mean"], p["W_log_var"], p["b_mean"], p["b_log_var"],
                                     eps_o.to(dt))
            loss = o.pow(2).mean()
            loss.backward()
            grads = {k: v.grad for k, v in p.items()}
            for i, sp in enumerate(specs):
                grads.update({f"flow_q.flows.{i}.{k}": v.grad for k, v in sp["params"].items()})
            return loss, grads

        best = float("inf")
        for _ in range(2):
            t1 = time.perf_counter()
            oracle_step(torch.float32)
            best = min(best, time.perf_counter() - t1)
        out["cpu_baseline"] = {"value": n / best, "unit": "rows/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle sample_z + MNFLinear.forward + torch.autograd backward of the mean-square "
                                         f"loss on the first {n} rows (initial parameters), best of 2, {best:.3f} s"}
        loss64, g64 = oracle_step(torch.float64)
        _, g32 = oracle_step(torch.float32)
        fresh.zero_grad()
        z, _ = fresh.sample_z(n, eps=eps_z.to(device), masks=[m.to(device) for m in masks])
        fresh.sample_z = lambda *a, **k: (z, None)
        loss_gpu = fresh.forward(x[:n].contiguous(), eps=eps_o.to(device)).pow(2).mean()
        loss_gpu.backward()
        worst, worst_name, worst_budget = 0.0, "", 0.0
        got = dict(fresh.named_parameters())
        for k, r64 in g64.items():
            if r64 is None or float(r64.abs().max()) == 0.0:
                continue
            scale = float(r64.abs().max())
            err = float((got[k].grad.cpu().double() - r64).abs().max()) / scale
            widening = 2.0 * float((g32[k].double() - r64).abs().max()) / scale
            if err - widening > worst - worst_budget or not worst_name:
                worst, worst_name, worst_budget = err, k, widening
        out["parity"] = {"rows": n, "reference": "oracle in float64 (same masks and noise)",
                         "worst_parameter_gradient_normwise_err": worst, "parameter": worst_name,
                         "fp32_oracle_vs_fp64_x2": worst_budget, "tolerance": 1e-5 + worst_budget,
                         "loss_gpu_vs_cpu_rel_err": abs(float(loss_gpu) - float(loss64)) / abs(float(loss64))}
    print(json.dumps(out))


def main_lenet(args, rank, world, device, dim, rows, desc) -> None:
    """The MNF example model's training step at the reference's batch size (128): a launch-bound step -- ~250 launches
    of a few microseconds each -- recorded once in a hipGraph (train.GraphedStep) and replayed.  What the library
    contributes: kl_div as one launch each way per layer, the one-row flows on the latency kernels, every parameter
    gradient added in place to one buffer, one Adam launch.  The convolutions are stock MIOpen."""
    if world != 1:
        raise SystemExit("--workload lenet measures one GPU")
    from torch import nn

    import torch_mnf_amd as amd

    torch.manual_seed(0)
    net = amd.MNFLeNet().to(device)  # (models/mnf_lenet.py:8-33)
    opt = amd.FusedAdam(amd.FlatParameters(net), lr=1e-3, capturable=True)
    gen = torch.Generator(device=device).manual_seed(1357)
    x = torch.rand(rows, 1, 28, 28, device=device, generator=gen)
    y = torch.randint(0, 10, (rows,), device=device, generator=gen)

    def loss_fn(xb, yb):
        return nn.functional.nll_loss(net(xb), yb) + net.kl_div() / 60000

    def eager():
        opt.zero_grad()
        loss = loss_fn(x, y)
        loss.backward()
        opt.step()
        return float(loss)

    first_loss = eager()
    for _ in range(2):
        eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        opt.zero_grad()
        loss_fn(x, y).backward()
        opt.step()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / 10 * 1e3
    step = amd.GraphedStep(opt, loss_fn, (x, y), model=net)
    t_prime = time.perf_counter()
    while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
        step(x, y)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(x, y)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    out = {
        "metric": f"images/s, {desc}", "value": rows * args.steps / elapsed, "unit": "images/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "batch": rows, "optimizer": "FusedAdam over FlatParameters (one buffer, one launch)",
                   "execution": "one hipGraph replay per step", "primed_ms": args.prime_ms},
        "eager_ms_per_step": eager_ms, "loss_first_step": first_loss, "loss_last_step": float(loss),
        "distributed": dist_info(1, "nccl", [elapsed], args.steps),
        "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "kernel": "(launch-bound: ~250 kernels of 3-75 us per step, no dominant one)", "avg_kernel_us": None},
    }
    print(json.dumps(out))


def main_train_c3(args, rank, world, device, dim, rows, desc) -> None:
    """Config 3's model on the training side of the bench contract: a step = one Adam step of 3 x [ActNorm, Glow,
    NSF_CL] on the resident batch (layer-by-layer forward keeping every intermediate, -mean log-prob, backward, Adam).
    The dominant kernels are the NSF_CL gradient pass's two launches (mnf_nsf_bwd_tile.hip: the conditioner as split
    MFMAs in both directions, 16 rows per wave, one launch per half-step; timed together, per layer).  The pass is bound
    by vector-instruction issue (the spline's derivative): the roofline object follows c3's "valu" convention, the HBM
    figures -- (12 d + 4) bytes per row per layer: x and grad_y in, grad_x out -- stay under their own names."""
    if world != 1:
        raise SystemExit("--workload c3t measures one GPU (data-parallel training would add a gradient all-reduce)")
    from torch_mnf_amd import flows as amd_flows

    model, layers = build_c3(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(rows, dim, device=device, generator=gen)  # resident in HBM before timing
    loss_box = [None]

    def step():
        opt.zero_grad()
        loss = -model.log_prob(x).mean()
        loss.backward()
        opt.step()
        loss_box[0] = loss

    step()
    torch.cuda.synchronize()
    first_loss = float(loss_box[0])
    t_prime = time.perf_counter()
    while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
        step()
        torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    amd_flows.bwd_kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    events, amd_flows.bwd_kernel_events = amd_flows.bwd_kernel_events, None
    kern_ms = [a.elapsed_time(b) for a, b in events]
    avg_s = sum(kern_ms) / len(kern_ms) / 1e3
    algo_bytes = (12 * dim + 4) * rows
    gbs = algo_bytes / avg_s / 1e9
    traffic, traffic_source = pmc_traffic("c3t")
    out = {
        "metric": f"samples/s, {desc}", "value": rows * args.steps / elapsed, "unit": "samples/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "rows_per_gpu": rows, "dim": dim, "layers": len(model.flows), "hidden": [8, 8, 8],
                   "K": 8, "tail_bound": 3.0, "optimizer": "torch.optim.Adam", "primed_ms": args.prime_ms,
                   "arithmetic": "fp32 (v_rcp / v_exp / v_log one-instruction forms in the spline gradient)",
                   "total_rows": rows},
        "distributed": dist_info(1, "nccl", [elapsed], args.steps),
        "loss_first_step": first_loss, "loss_last_step": float(loss_box[0]),
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, **physical(traffic, avg_s),
                     "kernel": "nsf_bwd_tile_kernel_16_8_8_inv_s0 + _s1 (+ the fixed-order reduction): one NSF_CL layer's "
                               "gradients = two launches, one per half-step; timed together; 3 layers per step",
                     "avg_kernel_us": avg_s * 1e6, "algorithmic_bytes_per_launch": algo_bytes,
                     "launches_timed": len(kern_ms), "launches_per_step": 3},
    }
    apply_valu_convention(out["roofline"], "c3t", avg_s)
    apply_transcendental_floor(out["roofline"], "c3t", rows, out["ms_per_step"])
    if not args.no_cpu_baseline:
        from oracle import flow_oracle as O

        n = 1 << 13
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        xs = x[:n].cpu()

        def specs_of(dtype):
            out_specs = []
            for l in layers:
                sp = {k: v for k, v in l.items() if k != "params"}
                sp["params"] = {k: (v.detach().clone().to(dtype).requires_grad_(True) if k != "P" else v.to(dtype))
                                for k, v in l["params"].items()}
                out_specs.append(sp)
            return out_specs

        best = float("inf")
        for _ in range(2):
            specs = specs_of(torch.float32)
            t1 = time.perf_counter()
            zs, ld = O.flow_stack(xs, specs, inverse=True)
            (-(ld + O.std_normal_log_prob(zs[-1])).mean()).backward()
            best = min(best, time.perf_counter() - t1)
        out["cpu_baseline"] = {"value": n / best, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle forward + torch.autograd backward of -mean log-prob on the first {n} rows "
                                         f"(initial parameters), best of 2, {best:.3f} s"}
        specs = specs_of(torch.float64)
        zs, ld = O.flow_stack(xs.double(), specs, inverse=True)
        loss_cpu = -(ld + O.std_normal_log_prob(zs[-1])).mean()
        loss_cpu.backward()
        fresh, _ = build_c3(device)
        loss_gpu = -fresh.log_prob(x[:n].contiguous()).mean()
        loss_gpu.backward()
        worst = 0.0
        for sp, f in zip(specs, fresh.flows):
            for name, prm in f.named_parameters():
                ref = sp["params"][name].grad
                if ref is None or prm.grad is None:
                    continue
                worst = max(worst, float((prm.grad.cpu().double().reshape(ref.shape) - ref).abs().max()
                                         / ref.abs().max().clamp_min(1e-30)))
        out["parity"] = {"rows": n, "reference": "oracle in float64", "worst_parameter_gradient_normwise_err": worst,
                         "tolerance": 1e-4,  # (tests/test_hip_autograd.py: the NSF_CL gradient bar)
                         "loss_gpu_vs_cpu_rel_err": abs(float(loss_gpu) - float(loss_cpu)) / abs(float(loss_cpu))}
    print(json.dumps(out))


# (key in `secondary`, workload, extra environment)
SECONDARY = (("c3", "c3", {}), ("c4", "c4", {}), ("c5", "c5", {}), ("c5b", "c5b", {}), ("c6", "c6", {}), ("c1", "c1", {}),
             ("c2_fp32", "c2", {"MNF_FP32_MFMA": "1"}),  # the headline on the strict fp32-MFMA stack kernel
             ("c2t", "c2t", {}), ("c3t", "c3t", {}), ("c5t", "c5t", {}), ("lenet", "lenet", {}))


def secondary_lines(args) -> dict:
    """The other configurations, driver-observable: after the headline's timed region (and outside it) each of them
    runs as a CHILD process -- `bench.py --workload X --steps 20 --warmup 3 --no-cpu-baseline` -- and its line is
    condensed to {ms_per_step, value, unit, kernel, avg_kernel_us, bound, frac, traffic, frac_physical} (+ the transcendental
    floor of c3 / c3t, + latency_us
    for c1).  (A child process, never an exec: this process has initialised the GPU.)  A workload that fails or
    overruns its time limit is reported as such."""
    import subprocess

    out = {}
    for key, w, extra_env in SECONDARY:
        # (MNF_BENCH_SECONDARY_STEPS: the test suite shortens the child runs)
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps",
               os.environ.get("MNF_BENCH_SECONDARY_STEPS", "20"), "--warmup", "3",
               "--no-cpu-baseline", "--no-secondary", "--prime-ms", str(args.prime_ms)]
        try:
            t0 = time.perf_counter()
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=90, env={**os.environ, **extra_env})
            line = next((l for l in reversed(res.stdout.splitlines()) if l.startswith("{")), None)
            if res.returncode != 0 or line is None:
                out[key] = {"error": (res.stderr or "no JSON line").strip().splitlines()[-1][:200]}
                continue
            d = json.loads(line)
            r = d.get("roofline", {})
            out[key] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                        "kernel": (r.get("kernel") or "")[:80], "avg_kernel_us": r.get("avg_kernel_us"),
                        "bound": r.get("bound"), "frac": r.get("frac"), "traffic": r.get("traffic"),
                        "frac_physical": r.get("frac_physical"), "wall_s": round(time.perf_counter() - t0, 1)}
            for k in ("transcendental_floor_ms", "frac_of_transcendental_floor"):  # (c3, c3t: the vector-issue-bound ones)
                if k in r:
                    out[key][k] = r[k]
            if "latency_us" in d:
                out[key]["latency_us"] = {k: v for k, v in d["latency_us"].items() if k != "note"}
            if extra_env:
                out[key]["env"] = extra_env
                out[key]["arithmetic"] = d.get("config", {}).get("arithmetic")
        except subprocess.TimeoutExpired:
            out[key] = {"error": "timed out after 90 s"}
    return out


def spawn_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without torchrun: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child process
    (one rank per GPU) and return its exit code.  Called before anything in this process touches the GPU."""
    import socket
    import subprocess

    with socket.socket() as sock:  # a free port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def rank_times(elapsed: float, world: int, device, backend: str) -> tuple[float, list[float]]:
    """(max over ranks, every rank's own time) of the timed region."""
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world == 1:
        return elapsed, [elapsed]
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    per_rank = [float(e.item()) for e in every]
    return max(per_rank), per_rank


def dist_info(world: int, backend: str, per_rank_s: list[float], steps: int) -> dict:
    """What the N > 1 line says about the ranks behind it (size of the process group the all-reduce ran on, the
    backend -- "nccl" is RCCL on ROCm --, every rank's own ms per step; `value` uses the slowest)."""
    return {"rccl_ranks": (dist.get_world_size() if world > 1 and backend == "nccl" else (1 if world == 1 else 0)),
            "ranks": dist.get_world_size() if world > 1 else 1,
            "collective_backend": backend if world > 1 else None,
            "ms_per_step_per_rank": [round(t / steps * 1e3, 4) for t in per_rank_s]}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prime-ms", type=float, default=60.0,
                    help="untimed device priming before the warm-up steps: the GPU leaves its idle power "
                         "state only after tens of ms of load, and the first ~25 passes run ~10 %% slower")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default (c2, one GPU) run only: skip the 20-step runs of the other configurations that are "
                         "attached to the line as `secondary`")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # No launcher around us: start the N ranks ourselves, as CHILD processes, before this process has
        # made any GPU call (a process that has initialised the GPU must never be replaced by another
        # program); rank 0's JSON line reaches our stdout through the launcher, its exit code becomes ours.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    # Self-test switches (not used by the driver): MNF_BENCH_BACKEND=gloo + MNF_BENCH_SHARE_GPU=1 run
    # the N > 1 control flow with every rank on GPU 0 of a one-GPU box.
    backend = os.environ.get("MNF_BENCH_BACKEND", "nccl")
    if os.environ.get("MNF_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import torch_mnf_amd as amd
    from torch_mnf_amd.dist import reduce_sum_count

    if not os.path.exists(amd.library_path()):  # normally prebuilt in-tree by __graft_entry__.build()
        if rank == 0:
            import __graft_entry__

            __graft_entry__.build()
        if world > 1:
            dist.barrier()

    dim, rows, desc = WORKLOADS[args.workload]
    if args.workload == "c5":
        return main_c5(args, rank, world, device, dim, rows, desc)
    if args.workload == "c5b":
        return main_c5(args, rank, world, device, dim, rows, desc, n_out=10)
    if args.workload == "c1":
        return main_c1(args, rank, world, device, dim, rows, desc)
    if args.workload == "c2t":
        return main_train(args, rank, world, device, dim, rows, desc)
    if args.workload == "c5t":
        return main_train_c5(args, rank, world, device, dim, rows, desc)
    if args.workload == "lenet":
        return main_lenet(args, rank, world, device, dim, rows, desc)
    if args.workload == "c3t":
        return main_train_c3(args, rank, world, device, dim, rows, desc)
    if args.workload in ("c3", "c3f"):
        model, layers = build_c3(device)
        if args.workload == "c3f":
            import torch_mnf_amd as _amd

            fl = list(model.flows)
            model = _amd.NormalizingFlowModel(model.base, [_amd.FusedSplineBlock(*fl[3 * i:3 * i + 3])
                                                           for i in range(3)]).to(device)
    else:
        model, layers = build_model(dim, device)
        if args.workload == "c2f":
            import torch_mnf_amd as _amd

            model = _amd.NormalizingFlowModel(model.base, [_amd.FusedAffineStack(list(model.flows))]).to(device)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(rows, dim, device=device, generator=gen)  # resident in HBM before timing
    n_layers = len(model.flows)

    pending = [None]

    def step():
        """One pass over the batch + the global mean.  world > 1: the 16-byte RCCL all-reduce of this pass is left
        in flight on RCCL's stream and collected at the start of the next pass, so it overlaps the next pass's
        kernel instead of sitting between two passes; finish() collects the last one inside the timed region."""
        lp, total = model.log_prob(x, return_sum=True)
        prev, pending[0] = pending[0], reduce_sum_count(total, rows, async_op=True)
        return prev.result() if prev is not None else None

    def finish():
        last, pending[0] = pending[0], None
        return last.result()

    with torch.no_grad():
        # setup: pack parameter images, let the caching allocator reach its steady state and bring
        # the device out of its idle power state (reported as config.primed_ms; not a timed step)
        step()  # first call: lazy library load, parameter packing, allocator growth
        finish()
        torch.cuda.synchronize()
        t_prime = time.perf_counter()
        primed = 0
        # (time-based, so ranks may run different numbers of passes: the priming pass is the local work only,
        #  WITHOUT the all-reduce of step() -- a collective here would deadlock ranks that disagree on the count)
        while (time.perf_counter() - t_prime) * 1e3 < args.prime_ms:
            model.log_prob(x, return_sum=True)
            torch.cuda.synchronize()
            primed += 1
        if world > 1:
            dist.barrier()
        # CPython's cyclic collector fires at a fixed allocation count, i.e. at the same layer of the
        # same step in every run, and a full collection of a process that has imported torch takes
        # ~70 ms: collect now and keep it off for the warm-up and timed steps.
        gc.collect()
        gc.disable()
        for _ in range(args.warmup):
            step()
        finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # (start, end) HIP events around every coupling kernel of the timed steps
        model.layer_events = None if os.environ.get("MNF_BENCH_NO_EVENTS") else []
        # layers of the inverse pass that get timed: all of them, except for c3 where only the dominant
        # kernel (the NSF_CL layer, first of every 3 in the inverse order) is
        # launch positions of one pass: (position in the inverse order, layers covered) -- a run of equal
        # AffineHalfFlow layers is ONE launch that still writes every intermediate
        launches = []
        if model.layer_events is not None:
            model.layer_event_pick = None
            step()
            finish()
            torch.cuda.synchronize()
            launches = [(i, span) for _, _, i, span in model.layer_events]
            model.layer_events = []
        else:
            launches = [(i, 1) for i in range(n_layers)]
        timed_layers = [i for i, _ in launches if args.workload != "c3" or i % 3 == 0]
        # one HIP event per step boundary (on the launch stream, no synchronisation): the per-step durations behind
        # `median_ms` / `min_ms` (SURVEY 8d asks for median + min next to the K-step wall clock)
        step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            # one layer per step carries the (start, end) marks, rotating: every mark costs a few us of
            # stream time, and ten of them per step would be ~4 % of the step they are measuring
            model.layer_event_pick = timed_layers[k % len(timed_layers)]
            step_marks[k].record()
            step()
        mean = finish()  # the last pass's mean: every pass's reduction completes inside the timed region
        step_marks[args.steps].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gc.enable()
        events, model.layer_events = model.layer_events, None
        if events is None:  # experiment mode: time the kernels in a separate pass instead
            model.layer_events = []
            for k in range(args.steps):
                model.layer_event_pick = timed_layers[k % len(timed_layers)]
                step()
            finish()
            torch.cuda.synchronize()
            events, model.layer_events = model.layer_events, None
        model.layer_event_pick = None

    # the other direction (sampling: z -> x), outside the timed region, for the record
    with torch.no_grad():
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(1, args.steps // 2)):
            xs_fwd, _ = model.forward(x)
        torch.cuda.synchronize()
        fwd_rate = rows * max(1, args.steps // 2) / (time.perf_counter() - t1)
        del xs_fwd

    elapsed, per_rank_s = rank_times(elapsed, world, device, backend)
    gpu_mean = float(mean.item())

    if rank == 0:
        by_layer, span_of = {}, {}
        for a, b, i, span in events:
            by_layer.setdefault(i, []).append(a.elapsed_time(b))
            span_of[i] = span
        span_dom = max(span_of.values())  # layers per launch of the dominant kernel
        if span_dom > 1:  # the fused run is the kernel the roofline is about
            by_layer = {i: v for i, v in by_layer.items() if span_of[i] == span_dom}
        per_layer_us = [round(1e3 * sum(v) / len(v), 1) for _, v in sorted(by_layer.items())]
        if os.environ.get("MNF_BENCH_DEBUG"):
            for i, v in sorted(by_layer.items()):
                print(f"layer {i} per step (us):", [round(t * 1e3) for t in v], file=sys.stderr)
        kern_ms = [t for v in by_layer.values() for t in v]
        avg_kernel_s = sum(kern_ms) / len(kern_ms) / 1e3
        step_ms = sorted(step_marks[k].elapsed_time(step_marks[k + 1]) for k in range(args.steps))
        kern_sorted = sorted(kern_ms)
        # per layer and row: read 4d, write 4d, log_det read+write; a launch that covers a run of
        # layers is priced at the algorithmic bytes of all of them, although it moves only
        # 4d (span + 1) + 8 bytes per row (each intermediate is written once and never re-read)
        algo_bytes = (8 * dim + 8) * rows * span_dom
        n_fused = 9 if args.workload == "c2f" else span_dom  # layers per launch (flop accounting)
        achieved = algo_bytes / avg_kernel_s / 1e9
        # (the committed PMC passes are of the default path; the layer-by-layer c2 kernel has its own file)
        traffic, traffic_source = pmc_traffic("c2_layer_by_layer" if args.workload == "c2" and span_dom == 1
                                              else args.workload)
        out = {
            "metric": "samples/s, 9xRNVP(AffineHalfFlow) d=64 batch=2^20 inverse+log-prob" if args.workload == "c2"
            else f"samples/s, {desc}",
            "value": world * rows * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "median_ms": step_ms[len(step_ms) // 2],   # per-step durations between HIP events on the launch stream
            "min_ms": step_ms[0],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "rows_per_gpu": rows, "dim": dim, "layers": n_layers,
                       "hidden": [8, 8, 8] if args.workload in ("c3", "c3f") else [24, 24, 24],
                       "intermediates": "all kept (reference API)", "primed_ms": args.prime_ms,
                       "primed_steps": primed, "arithmetic": ARITHMETIC, "total_rows": world * rows},
            "distributed": dist_info(world, backend, per_rank_s, args.steps),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # `achieved` / `frac` price the launch at SURVEY 8(d)'s ALGORITHMIC bytes: (8d + 8) per row per LAYER,
                # times the layers one launch covers.  A fused run moves fewer bytes than that (every intermediate is
                # written once, none is re-read): `traffic` (PMC) and `frac_physical` say what crosses the HBM
                # interface, so `frac` can exceed the physical utilisation -- never mix the two.
                "traffic": traffic,
                "traffic_source": traffic_source,
                **physical(traffic, avg_kernel_s),
                "kernel": ("nsf_mfma_kernel<16,8,8,inverse,block> ([NSF_CL, Glow, ActNorm] inverse in one launch, both "
                           "intermediates written)" if args.workload == "c3" and span_dom > 1 else
                           "nsf_mfma_kernel<16,8,8,inverse>" if args.workload == "c3" else "fused actnorm+glow+nsf_cl kernel (inverse)"
                           if args.workload == "c3f" else f"{AHF_KERNEL.replace('_kernel', '_stack_kernel')}<32,24,inverse> (9 layers per launch)"
                           if args.workload == "c2f" else
                           f"{AHF_KERNEL.replace('_kernel', '_stack_kernel')}<{dim // 2},24,inverse> ({span_dom} layers per launch, "
                           "every intermediate written)" if span_dom > 1 else
                           "ahf_rt_kernel<4,1,8,resident> (run-time-shaped: csrc/mnf_ahf_rt.hip; one layer per launch)"
                           if args.workload == "c6" else f"{AHF_KERNEL}<{dim // 2},24,inverse>"),
                "avg_kernel_us": avg_kernel_s * 1e6,
                "median_kernel_us": kern_sorted[len(kern_sorted) // 2] * 1e3,
                "min_kernel_us": kern_sorted[0] * 1e3,
                "algorithmic_bytes_per_launch": algo_bytes,
                "launches_timed": len(kern_ms),
                "layers_per_launch": span_dom,
                "bytes_moved_per_launch_by_design": (4 * dim * (span_dom + 1) + 8) * rows if span_dom > 1 else algo_bytes,
                "per_launch_us": [round(v, 1) for v in per_layer_us],
                "frac_of_achievable_6300": achieved / 6300.0,
                "fp32_tflops": (12800 if args.workload in ("c3", "c3f") else
                                n_fused * 2 * 2 * (2 * (dim // 2) * 24 + 2 * 24 * 24)) * rows / avg_kernel_s / 1e12,
            },
            "mean_log_prob": gpu_mean,
            "forward_direction_samples_per_s_per_gpu": fwd_rate,
        }
        if args.workload in ("c3", "c3f"):
            # the spline kernel is VALU / transcendental bound by construction (264 B per row against
            # ~2,300 vector instructions per 16 rows per layer): the roofline it is priced against is the vector
            # ISSUE rate; the HBM figures stay in the object under their own names.
            apply_valu_convention(out["roofline"], "c3", avg_kernel_s)
            apply_transcendental_floor(out["roofline"], "c3", rows, out["ms_per_step"])
        if world == 1 and not args.no_cpu_baseline:
            info, cpu_mean, n = cpu_baseline(layers, dim, x)
            with torch.no_grad():
                _, tot = model.log_prob(x[:n].contiguous(), return_sum=True)
            gpu_sample_mean = float(tot.item()) / n
            out["cpu_baseline"] = info
            out["parity"] = {"rows": n, "mean_log_prob_gpu": gpu_sample_mean, "mean_log_prob_cpu": cpu_mean,
                             "rel_err": abs(gpu_sample_mean - cpu_mean) / abs(cpu_mean), "tolerance": 1e-5}
            out["speedup_vs_cpu"] = out["value"] / info["value"]
        if world == 1 and args.workload == "c2" and not args.no_secondary:
            out["secondary"] = secondary_lines(args)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
