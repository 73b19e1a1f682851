#!/usr/bin/env python
"""bench.py — BASELINE.json metric on BASELINE.json configs[1]:
HiFi-GAN inference, batch = 64 utterances x 8 s (800 mel frames, hop 240 @ 24 kHz -> 192 064 samples each),
fp32, on N MI355X GPUs (one process per GPU; utterance shards, no data-path collective -> weak scaling).

A "step" is one generator forward over one resident batch of synthetic mels (inputs already in HBM).
Prints ONE JSON line on rank 0 (contract in the task prompt) with `roofline` and `cpu_baseline` objects.

    python bench.py                       # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(h, sd, budget_s=12.0):
    """Oracle (torch fp32 CPU restatement of the reference generator) on the host cores, bounded sample."""
    import torch
    from oracle import hifigan_ref as R
    ncpu = os.cpu_count() or 1
    w = R.fold_state_dict(sd)
    # pick the thread count that is actually fastest on this host (256 logical CPUs oversubscribe small convs)
    probe = R.synthetic_mel(1, 40, seed=1)
    best, cores = None, 1
    with torch.no_grad():
        for th in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
            torch.set_num_threads(th)
            R.generator_forward(w, h, probe)
            t0 = time.perf_counter()
            R.generator_forward(w, h, probe)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, cores = dt, th
            if dt > 3.0:
                break
    torch.set_num_threads(cores)
    B, T = 2, 300  # 2 utterances x 3 s (BASELINE configs[0] shape, doubled)
    mel = R.synthetic_mel(B, T, seed=1234)
    with torch.no_grad():
        n, t0 = 0, time.perf_counter()
        while True:
            out = R.generator_forward(w, h, mel)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 200:
                break
    samples = n * out.shape[0] * out.shape[2]
    return {'value': samples / el, 'unit': 'audio samples/s', 'cores': cores, 'kind': 'port',
            'sample': '%d x oracle generator_forward(B=%d, T=%d frames = 3 s) in %.1f s, torch CPU fp32, %d threads'
                      % (n, B, T, el, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=800, help='mel frames per utterance (800 = 8 s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from oracle import hifigan_ref as R  # synthetic weights/inputs + cpu_baseline leg only
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    h = dict(R.CONFIG_V1)
    sd = R.synthetic_state_dict(h, seed=1234)
    g = Generator(AttrDict(h))
    g.load_state_dict(sd)
    g = g.to(dev).eval()
    B, T = args.batch, args.frames
    # each rank owns its own shard of utterances (distinct seeds), resident in HBM before the timed region
    mel = R.synthetic_mel(B, T, seed=1234 + rank).to(dev)
    Lout = g.out_len(T)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # setup (not a step): the first forward packs + uploads the weights and allocates the workspace; a second one
        # brings the clocks up so that a small --warmup does not time the power ramp
        for _ in range(2):
            out = g(mel)
        barrier()
        for _ in range(args.warmup):
            out = g(mel)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            out = g(mel)
        ev1.record()
        barrier()
        elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels run on (torch current stream)
    assert bool(torch.isfinite(out).all())

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        samples_per_step = world * B * Lout
        value = samples_per_step * args.steps / elapsed
        flops_step = g.algorithmic_flops(B, T)  # per GPU, per step (1 168 559 FLOP/sample)
        achieved = flops_step * args.steps / (dev_ms * 1e-3) / 1e12   # ALGORITHMIC TFLOP/s (2 x MAC of the fp32 model)
        alg_bytes = B * (80 * T * 4 + Lout * 4)  # compulsory: mel in + wav out (5.33 B/sample)
        precision = g._precision
        if precision == 'f16x3':
            # split precision: every algorithmic product is three fp16 MFMA products (hi*hi + hi*lo + lo*hi, fp32 accumulate),
            # so the MFMA ceiling for fp32-accurate results is the dense f16 peak / 3
            peak, kname = PEAK_F16_MFMA_TFLOPS / 3.0, 'conv_f16x3_kernel + respair32_f16x3_kernel (v_mfma_f32_32x32x16_f16 x3 split-precision implicit-GEMM conv; all launches of one forward)'
            executed = 3.0 * achieved
        else:
            peak, kname = PEAK_FP32_MFMA_TFLOPS, 'conv_mfma_kernel (v_mfma_f32_32x32x2_f32 implicit-GEMM conv; all launches of one forward)'
            executed = achieved
        # HBM traffic of ONE forward of this exact workload, from the rocprofv3 PMC passes committed in
        # profiles/r01_bench_f16x3_hbm_pmc.csv (FETCH_SIZE 64.5 GB + WRITE_SIZE 62.1 GB, separate passes, KiB units; the
        # kernels stage with 4 B/lane loads, for which the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md — x2 for
        # 16 B/lane streams — does not apply).  Only reported for the configuration it was measured on.
        traffic = 126.6e9 if (precision == 'f16x3' and B == 64 and T == 800) else None
        res = {
            'metric': 'audio samples/sec (HiFi-GAN vocoder inference)', 'value': value, 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (split fp16 hi/lo x3 MFMA, fp32 accumulate)' if precision == 'f16x3' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'HiFi-GAN V1 generator inference, batch=%d utterances x %.1f s per GPU, 24 kHz, '
                                   '80-bin mel, config_v1 [5,3,4,4]' % (B, T * 240 / 24000.0),
                       'global_batch': world * B, 'frames': T, 'samples_per_utt': Lout,
                       'parallelism': 'utterance shards, no collective', 'precision': precision},
            'rtf_24k': value / world / 24000.0, 'rtf_22k05': value / world / 22050.0,
            'per_gpu_samples_s': value / world,
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'bytes per step (PMC, profiles/r01_bench_f16x3_hbm_pmc.csv)',
                         'kernel': kname,
                         'flops_per_step': flops_step, 'device_ms_per_step': dev_ms / args.steps,
                         'mfma_executed_tflops': executed, 'mfma_dense_peak_tflops': PEAK_F16_MFMA_TFLOPS if precision == 'f16x3' else PEAK_FP32_MFMA_TFLOPS,
                         'x_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'hbm_compulsory_GBs': alg_bytes * args.steps / (dev_ms * 1e-3) / 1e9,
                         'hbm_frac_compulsory': alg_bytes * args.steps / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:   # reported baseline: rank 0 at N=1 only
            res['cpu_baseline'] = cpu_baseline(h, sd)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
