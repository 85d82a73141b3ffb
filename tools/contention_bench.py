#!/usr/bin/env python3
"""What a kernel of another stream that holds X CUs (a collective during a data-parallel backward) does to the step, and what
VAW_P8_RESERVE_CUS buys back -- on ONE GPU, with a stand-in: `vaw_debug_cu_hog` parks X workgroups that each take a whole CU
on a side stream for the duration of every step.
    python tools/contention_bench.py [--hog 0,16,32] [--steps 20]          (run once per VAW_P8_RESERVE_CUS setting)"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vaw_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hog", default="0,16,32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    wl = bench.WORKLOADS["dit_b4"]
    dev = torch.device("cuda", 0)
    args = bench.workload_args(wl, parallel=False, amp=True, hip_graph=False)
    model, ema_model = bench.build(vaw_amd, wl, args, dev, 0)
    opt = vaw_amd.FusedAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=vaw_amd.get_lr_lambda(args))
    diff = vaw_amd.GaussianDiffusion(args=args, betas=vaw_amd.get_named_beta_schedule("cosine", 1000), model_mean_type=vaw_amd.ModelMeanType.EPSILON,
                                     model_var_type=vaw_amd.ModelVarType.FIXED_LARGE, loss_type=vaw_amd.LossType.MSE, rescale_timesteps=True)
    loader = bench._Loader(bench.synth_batches(a.batch, 4, dev, 123, wl))
    tr = vaw_amd.Trainer(args, dev, model, ema_model, opt, sched, diff, loader)
    side = torch.cuda.Stream()
    lib = vaw_amd.lib()
    for s in range(5):
        tr.train_step(s)
    torch.cuda.synchronize()
    for hog in (int(v) for v in a.hog.split(",")):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(a.steps):
            if hog:      # one hog launch per step, long enough to cover it; the next one queues behind it on the side stream
                vaw_amd._lib.check(lib.vaw_debug_cu_hog(hog, 14000, side.cuda_stream), "cu_hog")
            tr.train_step(10 + s)
        torch.cuda.current_stream().synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        torch.cuda.synchronize()
        print(f"reserve {os.environ.get('VAW_P8_RESERVE_CUS', '0'):>3}  hog {hog:3d} CUs: {ms:.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
