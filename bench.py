#!/usr/bin/env python
"""Headline benchmark: zoo ResNet-50 ComputationGraph training throughput (images/sec, whole job).

Config = BASELINE.json: ResNet-50 zoo ComputationGraph (reference ResNet50.java definition,
RmsProp + l1/l2 as in the zoo), batch 256 per GPU, synthetic 224x224 data (BenchmarkDataSetIterator
semantics: one random batch replayed), random-init weights, bf16 compute with fp32 master params.
Weak scaling: per-GPU batch fixed. Launch for N>1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
"""
import argparse
import math
import json
import os
import sys
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU minibatch")
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="disable HIP-graph capture of the iteration")
    ap.add_argument("--image", type=int, default=224)
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON result of rank 0): RCCL prints a version banner to fd 1 when its first
    # communicator comes up, so fd 1 is pointed at stderr for the duration of the run and the result goes to the saved fd.
    if os.environ.get("DL4J_AMD_BENCH_DUMP_AFTER"):          # debugging aid: all thread stacks to stderr after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DL4J_AMD_BENCH_DUMP_AFTER"]), exit=True)
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    from deeplearning4j_amd.parallel import comm
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.datasets.iterators import BenchmarkDataSetIterator
    from deeplearning4j_amd.parallel.parallel_wrapper import ParallelWrapper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU", file=sys.stderr)
        sys.exit(2)
    force_dp = os.environ.get("DL4J_AMD_FORCE_COLLECTIVES") == "1"      # 1-GPU rehearsal of the DP code path (debug only)
    if world > 1 or force_dp:
        # DL4J_AMD_DP_BACKEND=gloo: rehearsal of the multi-rank code path (capture cuts, bucket launches, parameter checksums) with
        # several ranks on ONE GPU, where RCCL refuses duplicate devices; the driver's runs use RCCL
        comm.init_process_group(os.environ.get("DL4J_AMD_DP_BACKEND", "nccl"))
    rank = comm.rank()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())

    model = ResNet50(numClasses=args.classes, inputShape=(3, args.image, args.image),
                     computeDataType="BFLOAT16" if args.dtype == "bf16" else None)
    net = model.init(device=dev)
    if not args.no_graph:
        net.enableGraphCapture(True)
    pw = None
    if world > 1 or force_dp:
        b = ParallelWrapper.Builder(net)
        if os.environ.get("DL4J_AMD_DP_OVERLAP") == "0":         # debugging aids for rehearsals: synchronous reduction / no watchdog
            b = b.overlap(False)
        if os.environ.get("DL4J_AMD_DP_WATCHDOG") == "0":
            b = b.watchdogTimeout(0)
        pw = b.build()

    it = BenchmarkDataSetIterator([args.batch, 3, args.image, args.image], args.classes, 1, device=dev, seed=1234 + rank)
    ds = it.next()
    x, y = ds.getFeatures(), ds.getLabels()

    for _ in range(args.warmup):
        net.fit(x, y)
    if pw is not None:
        pw.buckets.bytes_launched = pw.buckets.collectives = 0
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.fit(x, y)
    comm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    score = net.score()
    if not math.isfinite(float(score)):
        # a step that produced a non-finite loss did not do the work being timed: refuse to report it
        raise RuntimeError(f"non-finite training score after {args.warmup + args.steps} steps: {score}")
    # verification block (outside the timed region): what the collectives saw and whether the replicas still agree bit for bit
    if pw is not None:
        comm_rep = pw.comm_report(steps=args.steps)
    else:
        comm_rep = {"backend": "none", "world_size_seen": 1, "buckets": 0, "bytes_per_step": 0, "collectives_per_step": 0,
                    "params_in_sync": True}
    if rank == 0:
        gb = args.batch * world
        ips = gb * args.steps / dt
        line = json.dumps({
            "metric": "images/sec (whole node) ResNet-50 ComputationGraph training",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"model": "ResNet-50 zoo ComputationGraph (DL4J ResNet50.java definition)", "global_batch": gb,
                       "per_gpu_batch": args.batch, "image": "3x%dx%d" % (args.image, args.image), "classes": args.classes,
                       "updater": "RmsProp(0.1,0.96,0.001)+l1 1e-7+l2 5e-5", "parallelism": "dp%d" % world,
                       "hip_graph": not args.no_graph, "final_score": round(float(score), 4)},
            "comm": comm_rep})
        sys.stdout.flush()
        os.write(result_fd, (line + "\n").encode())
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
