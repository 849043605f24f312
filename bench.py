#!/usr/bin/env python3
"""bench.py — CelebA-128 cold-diffusion training throughput (+ 200-step sampling latency) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 3, the single-GPU CelebA-128 case):
  Unet(dim=64, dim_mults=(1,2,4,8), channels=3) at 128x128, denoising GaussianDiffusion(T=200,
  x0_step_down), synthetic 8-bit-quantised images, random-init weights.
A *step* is one optimizer step exactly as Trainer.train() does it (DEBLUR:1188-1204): 2 accumulation
micro-steps of 32 images (q_sample -> UNet fwd -> L1 -> bwd) + Adam (+ EMA every 10th step); with
N > 1 each rank does the same work on its own shard and gradients are all-reduced over RCCL
(weak scaling).  `value` = images/s over all ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


_T0 = time.perf_counter()
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
UNET128_FWD_GFLOP = 67.41          # SURVEY.md §8(d), per image
TIMESTEPS = 200


def build_workload(args, device):
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    import contextlib
    import io
    torch.manual_seed(123457)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(device)
    diffusion = GaussianDiffusion(model, image_size=128, channels=3, timesteps=TIMESTEPS, loss_type='l1',
                                  sampling_routine='x0_step_down').to(device)
    trainer = Trainer(diffusion, None, image_size=128, train_batch_size=args.batch, train_lr=2e-5, train_num_steps=10 ** 9,
                      gradient_accumulate_every=args.accum, ema_decay=0.995, fp16=False, dataset='synthetic',
                      results_folder=os.path.join(REPO, "gpurun_out", "bench_results"))
    trainer.quiet = True
    return model, diffusion, trainer


class GemmTimer:
    """HIP-event timing (on the launch stream) of every MFMA GEMM launch of the step with its
    ALGORITHMIC FLOPs (zero-padded taps / channels are not counted):
      conv_igemm_sp   cdf_conv_gemm_bf16(x)   dense conv fwd + dgrad, bf16x3 split-precision MFMA
      conv_wgrad_sp   cdf_conv_wgrad_bf16(x)  weight gradients of the wide layers, bf16x3 MFMA
      conv_igemm      cdf_conv_gemm           exact-fp32 MFMA (short-K 1x1 convs, K=32 attention GEMMs, linears)
      conv_wgrad      cdf_conv_wgrad          weight gradients of the thin layers, exact-fp32 MFMA
    """
    PEAK = {"conv_igemm_sp": 2500.0, "conv_wgrad_sp": 2500.0, "conv_igemm": 157.3, "conv_wgrad": 157.3}   # dense TFLOP/s, MI355X_MICROARCH.md
    _SP = "bf16 MFMA, 3 MFMAs per product (hi/lo split operands, fp32 accumulate)"
    DTYPE = {"conv_igemm_sp": _SP, "conv_wgrad_sp": _SP, "conv_igemm": "f32 MFMA", "conv_wgrad": "f32 MFMA"}
    ENTRY = {"cdf_conv_gemm": "conv_igemm", "cdf_conv_gemm_bf16": "conv_igemm_sp", "cdf_conv_gemm_bf16x": "conv_igemm_sp",
             "cdf_conv_wgrad": "conv_wgrad", "cdf_conv_wgrad_bf16": "conv_wgrad_sp", "cdf_conv_wgrad_bf16x": "conv_wgrad_sp"}

    def __init__(self, lib):
        self.lib, self.enabled, self.records, self.taps = lib, False, {k: [] for k in self.PEAK}, {}
        self.orig = {}
        for entry in self.ENTRY:
            self.orig[entry] = getattr(lib, entry)
            setattr(lib, entry, (lambda e: lambda *a: self._call(e, a))(entry))

    def _ntaps(self, desc, nphase):
        key = id(desc)
        if key not in self.taps:
            tot, i = 0, 0
            for _ in range(nphase):
                tot += desc[i + 2]
                i += 3 + 3 * desc[i + 2]
            self.taps[key] = (tot, desc)          # keep desc alive so id() stays unique
        return self.taps[key][0]

    def _flops(self, kind, a):
        if kind == "cdf_conv_gemm_bf16x":   # (xhi,xlo,ldx,zero,whi,wlo,ldk,y,ldy,B,H,W,Cin,OH,OW,Cout,QH,QW,os,is,nphase,desc,...)
            return 2.0 * a[9] * a[16] * a[17] * self._ntaps(a[21], a[20]) * a[12] * a[15]
        if kind == "cdf_conv_wgrad_bf16x":  # (ahi,alo,lda,bhi,blo,ldb,zero,ws,ldo,B,QH,QW,HA,WA,sa,HB,WB,sb,CA,CB,ntaps,...)
            return 2.0 * a[9] * a[10] * a[11] * a[18] * a[19] * a[20]
        if kind == "cdf_conv_wgrad_bf16":   # (xa,lda,xb,ldb,ws,ldo,B,QH,QW,HA,WA,sa,HB,WB,sb,CA,CB,ntaps,...)
            return 2.0 * a[6] * a[7] * a[8] * a[15] * a[16] * a[17]
        if kind == "cdf_conv_gemm":       # (x,ldx,w,ldw,y,ldy,B,H,W,Cin,OH,OW,Cout,QH,QW,os,is,nphase,desc,...,batch@32,...,batch2@36)
            return 2.0 * a[6] * a[13] * a[14] * self._ntaps(a[18], a[17]) * a[9] * a[12] * a[32] * a[36]
        if kind == "cdf_conv_gemm_bf16":    # (x,ldx,whi,wlo,ldk,y,ldy,B,H,W,Cin,OH,OW,Cout,QH,QW,os,is,nphase,desc,...)
            return 2.0 * a[7] * a[14] * a[15] * self._ntaps(a[19], a[18]) * a[10] * a[13]
        # conv_wgrad: (xa,lda,xb,ldb,ws,ldo,B,QH,QW,HA,WA,sa,HB,WB,sb,CA,CB,ntaps,desc,nsplit,batch,...)
        return 2.0 * a[6] * a[7] * a[8] * a[15] * a[16] * a[17] * a[20]

    def _call(self, entry, a):
        if not self.enabled:
            return self.orig[entry](*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = self.orig[entry](*a)
        e1.record()
        self.records[self.ENTRY[entry]].append((e0, e1, self._flops(entry, a), (entry,) + tuple(x for x in a[6:21] if isinstance(x, int) and x < 1 << 20)))
        return r

    def dump_shapes(self, steps):
        """Per-shape time table (stderr) — which GEMMs the step spends its time in."""
        agg = {}
        for recs in self.records.values():
            for e0, e1, f, key in recs:
                t, n, fl = agg.get(key, (0.0, 0, 0.0))
                agg[key] = (t + e0.elapsed_time(e1), n + 1, fl + f)
        for key, (t, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
            log("%8.3f ms/step %4d launches %7.1f TF  %s" % (t / steps, n // steps, fl / (t * 1e-3) / 1e12 if t > 0 else 0, key))

    def summary(self, steps, elapsed_s):
        out = {}
        for kind, recs in self.records.items():
            if not recs:
                continue
            ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in recs)
            fl = sum(r[2] for r in recs)
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out[kind] = {"arithmetic": self.DTYPE[kind], "achieved": round(tf, 2), "peak": self.PEAK[kind], "unit": "TFLOP/s",
                         "frac": round(tf / self.PEAK[kind], 4), "launches_per_step": len(recs) // max(1, steps),
                         "avg_launch_ms": round(ms / len(recs), 4), "algorithmic_gflop_per_step": round(fl / max(1, steps) / 1e9, 1),
                         "share_of_step": round(ms / (1000 * elapsed_s), 3)}
        return out


def cpu_baseline(args):
    """The CPU oracle (port of the reference's PyTorch path) on the host cores: the same optimizer
    step on a bounded sample (2 micro-steps x 2 images)."""
    from oracle import cold_oracle as O
    from colddiff.unet import Unet
    import contextlib
    import io
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncores = max(1, min(ncores, 64))
    torch.set_num_threads(ncores)
    torch.manual_seed(123457)
    with contextlib.redirect_stdout(io.StringIO()):
        sd = {k: v.clone() for k, v in Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).state_dict().items()}
    ca, cb = O.cosine_tables(TIMESTEPS)
    Bc = 2

    def loss_of_batch(params, x, eps, t):
        return O.loss_fn(x, O.unet_forward(params, O.noise_q_sample(x, eps, t, ca, cb), t))

    tr = O.OracleTrainer(sd, loss_of_batch, lr=2e-5, accumulate=args.accum)
    g = torch.Generator().manual_seed(123457)

    def batches():
        out = []
        for _ in range(args.accum):
            x = torch.randint(0, 256, (Bc, 3, 128, 128), generator=g).float() / 255 * 2 - 1
            out.append((x, torch.randn(Bc, 3, 128, 128, generator=g), torch.randint(0, TIMESTEPS, (Bc,), generator=g)))
        return out

    t0 = time.perf_counter()
    tr.train_step(batches())                     # warm-up
    warm = time.perf_counter() - t0
    log(f"cpu baseline warm-up step: {warm:.1f}s on {ncores} threads")
    n = 2 if warm < 12 else 1                    # keep the whole leg within ~30 s
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(batches())
    dt = (time.perf_counter() - t0) / n
    return {"value": round(Bc * args.accum / dt, 3), "unit": "img/s", "cores": ncores, "kind": "port",
            "sample": f"oracle/cold_oracle.py (PyTorch CPU fp32) optimizer step, {args.accum} micro-steps x {Bc} images at 128x128, "
                      f"{n} timed steps after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per micro-step per GPU (reference scripts: 32)")
    ap.add_argument("--accum", type=int, default=2, help="gradient_accumulate_every (reference scripts: 2)")
    ap.add_argument("--sample-batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sample", action="store_true")
    args = ap.parse_args()

    from colddiff import parallel, runtime
    world = parallel.world_size()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    if world > 1:
        parallel.init_distributed("nccl")
    device = torch.device("cuda", parallel.local_rank())
    torch.cuda.set_device(device)

    log(f"building workload on {device} (world {world})")
    model, diffusion, trainer = build_workload(args, device)
    timer = GemmTimer(runtime.lib())
    use_timer = os.environ.get("CDF_BENCH_NOTIMER", "0") != "1"
    log("workload built")

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.train_step()
        trainer.step += 1
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step()
        trainer.step += 1
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed {args.steps} steps: {elapsed:.3f}s")
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()

    # Roofline pass: the SAME K steps again, now with a HIP-event pair around every MFMA GEMM launch (on the launch
    # stream).  Kept out of the timed region above because ~1000 event records per step cost 3 % of throughput; the
    # instrumented steps' own wall time is reported next to the kernel figures.
    elapsed_instr = None
    if use_timer:
        timer.enabled = True
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_step()
            trainer.step += 1
        barrier()
        elapsed_instr = time.perf_counter() - t1
        timer.enabled = False
        log(f"instrumented {args.steps} steps: {elapsed_instr:.3f}s")

    imgs_per_step = args.batch * args.accum * world
    value = imgs_per_step * args.steps / elapsed
    kernels = timer.summary(args.steps, elapsed_instr if elapsed_instr else elapsed)
    if os.environ.get("CDF_BENCH_SHAPES"):
        timer.dump_shapes(args.steps)

    out = {
        "metric": "unet_train_imgs_per_sec", "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if runtime.precision == "f32" else ("f32 storage/accumulate; dense-conv GEMM operands " + runtime.precision),
        "data": "synthetic",
        "config": {"workload": "CelebA-128 denoising cold diffusion (BASELINE config 3): Unet(dim=64,(1,2,4,8),ch=3) @128x128, T=200, "
                               "optimizer step = 2 micro-steps x 32 img + Adam + EMA/10", "per_gpu_batch": args.batch,
                   "gradient_accumulate_every": args.accum, "global_images_per_step": imgs_per_step,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
    }
    if parallel.rank() == 0:
        dom = max(kernels, key=lambda k: kernels[k]["share_of_step"]) if kernels else None
        if dom:
            d = kernels[dom]
            out["roofline"] = {"bound": "mfma", "kernel": dom, "arithmetic": d["arithmetic"], "achieved": d["achieved"], "peak": d["peak"],
                               "unit": "TFLOP/s", "frac": d["frac"], "traffic": None, "launches_per_step": d["launches_per_step"],
                               "avg_launch_ms": d["avg_launch_ms"], "algorithmic_gflop_per_step": d["algorithmic_gflop_per_step"],
                               "share_of_step": d["share_of_step"],
                               "measured": "HIP events around every launch of %d further steps run right after the timed region "
                                           "(%.1f ms/step with the ~1000 event records per step, %.1f ms/step without)"
                                           % (args.steps, 1000 * elapsed_instr / args.steps, 1000 * elapsed / args.steps)}
            # HBM traffic of that kernel group from the committed rocprofv3 --pmc passes over this same command
            # (tools/pmc_traffic.py; bench.py cannot run the profiler on itself)
            tpath = os.path.join(REPO, "profiles", "round1_pmc_traffic.json")
            if os.path.exists(tpath) and dom == "conv_igemm_sp":
                g = json.load(open(tpath)).get("conv_igemm_sp")
                if g:
                    out["roofline"]["traffic"] = round(g["hbm_bytes_per_launch"])
                    out["roofline"]["traffic_unit"] = "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/round1_pmc_traffic.json)"
        out["gemm_kernels"] = kernels
        # whole-step view: 3 x F_fwd per image against the same MFMA peak
        step_tflops = 3 * UNET128_FWD_GFLOP * args.batch * args.accum * args.steps / elapsed / 1e3
        out["step_algorithmic_tflops"] = round(step_tflops, 2)
        out["step_frac_of_f32_mfma_peak"] = round(step_tflops / PEAK_F32_MFMA_TFLOPS, 4)
        if world == 1 and not args.no_sample:
            with torch.no_grad():
                noise = torch.randn(args.sample_batch, 3, 128, 128, device=device)
                trainer.ema_core.gen_sample(batch_size=args.sample_batch, img=noise, t=2)      # warm-up
                torch.cuda.synchronize()
                ts = time.perf_counter()
                trainer.ema_core.gen_sample(batch_size=args.sample_batch, img=noise)
                torch.cuda.synchronize()
                out["sample_ms_per_img_200step"] = round(1000 * (time.perf_counter() - ts) / args.sample_batch, 2)
                log(f"200-step gen_sample of {args.sample_batch} images: {time.perf_counter() - ts:.2f}s")
                out["sample_batch"] = args.sample_batch
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
