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
TRAFFIC_CSV = os.path.join(ROOT, 'profiles', 'r06_bench_hbm_pmc.csv')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this workload


def measured_traffic(precision, B, T):
    """HBM bytes of ONE forward from the committed PMC summary (profiles/rNN_bench_hbm_pmc.csv: separate FETCH_SIZE and
    WRITE_SIZE passes over `bench.py --steps 1`, summed over all kernels of a forward, gfx950 corrections of
    MI355X_MICROARCH.md applied in the file's `bytes_per_forward` row).  None when the file does not describe this
    configuration — the number is never a literal in this script."""
    try:
        rows = [l.strip() for l in open(TRAFFIC_CSV) if l.strip() and not l.startswith('#')]
        kv = dict(l.split(',', 1) for l in rows if l.count(',') == 1)
        if kv.get('precision') == precision and int(kv.get('batch', -1)) == B and int(kv.get('frames', -1)) == T:
            return float(kv['bytes_per_forward'])
    except Exception:
        pass
    return None


def live_traffic(precision, B, T, timeout_s=150):
    """HBM bytes of ONE forward measured NOW: two child runs of this script (`--steps 1 --warmup 0`: 3 identical forwards, TTSC_HIFIGAN_CALIBRATE=0) under
    `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and `--pmc SQ_INSTS_MFMA` (separate passes, --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes), counters summed
    over all ttsc:: kernels, divided by 3, FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads: an upper bound), WRITE_SIZE checked against
    the launch that writes exactly the waveform.  The recipe of tools/profile_r06.sh + tools/hbm_from_pmc.py, run after every timed region of this process.
    Returns (bytes or None, detail): any failure (no rocprofv3, a pass that does not finish in `timeout_s`, an unreadable database) returns None and the caller
    falls back to the committed summary."""
    import glob
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rp):
        return None, 'rocprofv3 not found'
    tmp = tempfile.mkdtemp(prefix='ttsc_pmc_', dir='/tmp')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(TMPDIR='/tmp', TTSC_HIFIGAN_CALIBRATE='0')
    tot, post = {}, {}
    try:
        for cn in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_MFMA'):
            d = os.path.join(tmp, cn)
            cmd = [rp, '--pmc', cn, '--kernel-trace', '-d', d, '-o', 'p', '--', sys.executable, os.path.abspath(__file__), '--no-extra', '--no-cpu-baseline',
                   '--steps', '1', '--warmup', '0', '--batch', str(B), '--frames', str(T)]
            pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)      # (exactly the process group started above)
                pr.wait()
                return None, '%s pass did not finish in %d s' % (cn, timeout_s)
            dbs = glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True)
            if pr.returncode != 0 or not dbs:
                return None, '%s pass failed (rc %s)' % (cn, pr.returncode)
            c = sqlite3.connect(dbs[0])
            rows = c.execute('select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name', (cn,)).fetchall()
            c.close()
            mine = [(k, v, n) for k, v, n in rows if 'ttsc::' in k and 'mfma_sustained_kernel' not in k]   # (the register-only probe loop is not part of a forward)
            if not mine:
                return None, '%s pass recorded no ttsc:: kernel' % cn
            tot[cn] = sum(v for _, v, _ in mine) * (1.0 if cn == 'SQ_INSTS_MFMA' else 1024.0) / 3.0           # byte counters in KiB; 3 forwards in the child run
            if cn == 'WRITE_SIZE':
                pk = [(v, n) for k, v, n in mine if 'rbchain_f16x3_kernel' in k and k.split('>')[0].rstrip().endswith('true, true')]
                if pk:
                    post = {'per_launch': pk[0][0] * 1024.0 / pk[0][1]}
    except Exception as e:
        return None, 'live PMC passes failed: %s' % str(e)[:120]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    Lout = T
    for u, k in ((5, 16), (3, 16), (4, 4), (4, 4)):
        Lout = (Lout - 1) * u - 2 * ((k - u) // 2) + k
    want = B * Lout * 4
    detail = {'fetch_size_raw_bytes': tot['FETCH_SIZE'], 'write_size_raw_bytes': tot['WRITE_SIZE'], 'sq_insts_mfma_per_forward': tot['SQ_INSTS_MFMA'],
              'write_calibration_ratio': (post['per_launch'] / want) if post else None,
              'recipe': 'bytes = 2 x FETCH_SIZE + WRITE_SIZE over all ttsc:: kernels of 3 forwards / 3 (KiB counters; gfx950: FETCH_SIZE reports half the bytes of wide reads)'}
    return 2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE'], detail


PMC_CSV = os.path.join(ROOT, 'profiles', 'r06_bench_pmc.csv')            # per-kernel counter sums of the same command (3 forwards)
UNFUSED_BYTES_PER_SAMPLE = (5.33 + 8 + 16 + 32) * 51 * 4                 # SURVEY.md §8d "unfused layer-boundary" model: 12.5 KB per output sample


def measured_mfma_insts():
    """SQ_INSTS_MFMA of ONE forward of the headline workload from the committed counter summary (all ttsc:: convolution kernels of
    `bench.py --steps 1 --warmup 0`: 3 forwards, divided by 3; the register-only probe loop excluded), or None"""
    try:
        import csv
        rows = [l for l in open(PMC_CSV) if not l.startswith('#')]
        tot = 0.0
        for r in csv.DictReader(rows):
            if 'mfma_sustained_kernel' in r['kernel'] or not r.get('SQ_INSTS_MFMA'):
                continue
            tot += float(r['SQ_INSTS_MFMA'])
        return tot / 3.0 if tot else None
    except Exception:
        return None


def self_check(g, mel, out, h, sd, tol=1e-4):
    """Refuse to report a time for wrong results: the head of utterance 0 (30 frames = 7 200 samples, beyond the ~4.9 k-sample
    receptive field) must equal the oracle run on a 60-frame prefix within the parity gate (1e-4 RMS)."""
    import torch
    from oracle import hifigan_ref as R
    T = min(60, mel.shape[2])
    ref = R.generator_forward(R.fold_state_dict(sd), h, mel[:1, :, :T].cpu())
    n = min(240 * 30, ref.shape[2]) if mel.shape[2] > T else ref.shape[2]
    d = out[0, 0, :n].detach().cpu() - ref[0, 0, :n]
    rms = float(d.pow(2).mean().sqrt())
    assert rms < tol, 'bench self-check failed: rms %.3e vs the oracle on utterance 0' % rms
    return rms


def time_forward(g, mel, steps, warmup, check='sync'):
    """(device ms per forward, last output) of `steps` forwards after `warmup`, HIP events on the launch stream.  check='deferred': the
    range guard's verdicts are collected at the next call / before the clock stops instead of by a stream synchronisation per call."""
    import torch
    with torch.no_grad():
        for _ in range(warmup):
            out = g(mel, check=check)
        g.finish_range_check()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            out = g(mel, check=check)
        g.finish_range_check()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out


def extra_legs(g, h, sd, rank_dev, R):
    """Secondary measurements the driver times with the same command: exact-fp32 arithmetic on the same
    workload, the reference API's own B=1 x 3 s case (BASELINE configs[0] shape on the GPU), WaveRNN decode (configs[2]), one
    Cubegan training step (configs[3] per-GPU share) and text features -> audio end to end (configs[4] per-GPU share)."""
    import numpy as np
    import torch
    legs = {}
    try:
        mel1 = R.synthetic_mel(1, 300, seed=77).to(rank_dev)
        ms, o1 = time_forward(g, mel1, 50, 20, check='deferred')   # 50 back-to-back calls: pipelined, guard verdicts collected in the timed region
        ms_sync, _ = time_forward(g, mel1, 50, 5)                   # ... and with the default per-call synchronisation (what a caller who reads the audio back sees)
        legs['single_utterance_3s'] = {'ms': ms, 'ms_with_per_call_sync': ms_sync, 'samples_per_s': o1.shape[2] / (ms * 1e-3),
                                       'rms_vs_oracle': self_check(g, mel1, o1, h, sd)}
    except Exception as e:   # a secondary leg must never take the headline number down
        legs['single_utterance_3s'] = {'error': str(e)[:200]}
    try:
        from oracle import wavernn_ref as WO
        from ttscube_amd.networks.modules import WaveRNN
        wsd = WO.synthetic_state_dict(H=512, num_layers=1, use_lowres=True, seed=5)
        net = WaveRNN(num_layers=1, layer_size=512, upsample=240, upsample_low=10, use_lowres=True, output='mulaw')
        net.load_state_dict({k: torch.from_numpy(v) for k, v in wsd.items()}, strict=True)
        net = net.to(rank_dev).eval()
        Bw, Tw = 256, 100   # C3 = 256 utterances x 100 frames = 24 000 autoregressive steps
        wm, wx = WO.synthetic_inputs(Bw, Tw, seed=6)
        X = {'mel': torch.from_numpy(wm), 'x_low': torch.from_numpy(wx)}
        net.decode({'mel': X['mel'][:, :4], 'x_low': X['x_low'][:, :96]}, mode='philox', seed=1)   # warm-up (weights packed, clocks up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, _, _ = net.decode(X, mode='philox', seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # oracle on utterances 0 and 255 over the WHOLE decode (the philox counters carry the utterance index: b_offset)
        from concurrent.futures import ThreadPoolExecutor
        chk = (0, Bw - 1)
        with ThreadPoolExecutor(len(chk)) as ex:
            refs = list(ex.map(lambda b: WO.decode(wsd, wm[b:b + 1], wx[b:b + 1], num_layers=1, H=512, mode=WO.MODE_PHILOX, seed=2,
                                                   b_offset=b)[0], chk))
        idx_h = idx.cpu().numpy()
        for b, ridx in zip(chk, refs):
            assert np.array_equal(idx_h[b], ridx[0]), 'WaveRNN indices of utterance %d differ from the oracle' % b
        legs['wavernn_decode_b256'] = {'us_per_step': dt / (Tw * 240) * 1e6, 'samples_per_s': Bw * Tw * 240 / dt, 'H': 512, 'layers': 1,
                                       'frames': Tw, 'steps': Tw * 240, 'output': 'mulaw', 'indices_bit_exact_vs_oracle': True,
                                       'oracle_checked': '%d utterances x %d steps' % (len(chk), Tw * 240), 'kernel': net.last_kernel}
        # roofline of the decode loop (1 139 712 MAC per sample and utterance, one layer H = 512): the chains run on the fp32 matrix
        # pipe (bit-exact contract), so the peak is the fp32 MFMA rate; the kernel is latency-bound by construction (one dependent step at a time)
        wr_tf = Bw * 2.0 * 1139712 / (dt / (Tw * 240)) / 1e12
        legs['wavernn_decode_b256']['roofline'] = {'bound': 'mfma', 'achieved': wr_tf, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                                   'frac': wr_tf / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
                                                   'note': 'per-step latency chain (two dependent products + three hand-offs); weights stay on chip, HBM traffic negligible'}
    except Exception as e:
        legs['wavernn_decode_b256'] = {'error': str(e)[:200]}
    # BASELINE configs[3] per-GPU share: full Cubegan training steps (no exchange at N = 1; `--mode train` runs them under RCCL) at b = 16 (the
    # reference script's default batch) and b = 128 (the per-GPU batch configs[3] names), through the class surface the reference's trainer drives
    for b_tr in (16, 128):
        tag = 'cubegan_training_step_b%d' % b_tr
        try:
            import random
            from ttscube_amd.io_utils.io_cubegan import CubeganCollate
            from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
            from ttscube_amd.networks.cubegan import Cubegan
            enc = synthetic_encodings()
            torch.manual_seed(1234)
            model = Cubegan(enc, conditioning=None, train=True).to(rank_dev)
            model.train()
            batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(b_tr, 777, min_ph=30, max_ph=50)))
            crop = random.Random(99)
            n_warm, n_timed = (4, 8) if b_tr <= 16 else (2, 3)   # (steps 0 / 1 lay out the optimizer arenas and the weight banks; steady state from step 2)
            for _ in range(n_warm):
                out = model.training_step(batch, 0, rng=crop)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_timed):
                out = model.training_step(batch, 0, rng=crop)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_timed
            tfl, _ = cubegan_step_conv_flops(b_tr)
            legs[tag] = {'ms_per_step': dt * 1e3, 'samples_per_s': b_tr * 12000 / dt, 'steps_timed': n_timed, 'steps_warmup': n_warm, 'losses': {k: round(float(v), 5) for k, v in out.items()},
                         'roofline': {'bound': 'mfma', 'achieved': tfl / dt / 1e12, 'peak': PEAK_F16_MFMA_TFLOPS / 3, 'unit': 'TFLOP/s',
                                      'frac': tfl / dt / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3), 'traffic': None, 'flops_per_step': tfl,
                                      'note': 'convolution FLOPs of the step (bench.py::cubegan_step_conv_flops) over the whole step time'}}
            model._optimizers = None
            del model
            torch.cuda.empty_cache()
        except Exception as e:
            legs[tag] = {'error': str(e)[:200]}
    try:   # BASELINE configs[4] per-GPU share: text features -> audio (Languasito2 + generator), 64 random sentences and one sentence
        import numpy as np
        from oracle import meldecoder_ref as MO   # synthetic weights only
        from ttscube_amd.networks.cubegan import Cubegan

        class _Enc:
            phon2int = {'p%d' % i: i for i in range(50)}
            speaker2int = {'s0': 0}
            max_pitch = 300
            max_duration = 12   # synthetic weights give ~uniform durations: ~6 frames per phoneme

        torch.manual_seed(0)
        tts = Cubegan(_Enc(), conditioning=None, train=False)
        esd = tts.state_dict()
        esd.update({'_languasito.' + k: v for k, v in MO.fill_state_dict(MO.named_shapes(tts._languasito), 5).items()})
        esd.update({'_generator.' + k: v for k, v in sd.items()})
        tts.load_state_dict(esd)
        tts = tts.to(rank_dev).eval()
        from ttscube_amd.io_utils.synthetic import synthetic_sentences
        n = 64
        xc, lens = synthetic_sentences(n, seed=1234)
        lsd = {k[len('_languasito.'):]: v for k, v in esd.items() if k.startswith('_languasito.')}
        wfold = R.fold_state_dict(sd)
        to16 = lambda a: np.asarray(a * 32767, dtype=np.int16)
        for tag, xx in (('e2e_64_sentences', xc), ('e2e_single_sentence', xc[:1, :lens[0]])):
            mk = lambda: {'x_char': torch.from_numpy(xx), 'x_speaker': torch.ones((xx.shape[0], 1), dtype=torch.long)}
            for _ in range(2):
                wav, wl = tts.inference(mk(), return_lengths=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):   # (deferred range guard: verdicts collected at the next call / below, no pipeline stall)
                wav, wl = tts.inference(mk(), return_lengths=True, check='deferred')
            tts._generator.finish_range_check()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            assert bool(torch.isfinite(wav).all())
            # never report a time for wrong audio: shortest + longest sentence against the oracle chain (meldecoder_ref -> hifigan_ref):
            # identical durations, <= 4 LSB int16 (1e-4 of full scale)
            worst = 0
            for b in sorted({int(np.argmin(lens[:xx.shape[0]])), int(np.argmax(lens[:xx.shape[0]]))}):
                with torch.no_grad():
                    cond, durs, _ = MO.languasito2_inference(lsd, torch.from_numpy(xc[b:b + 1, :lens[b]]), torch.tensor([[1]]), _Enc.max_pitch)
                    ref = R.generator_forward(wfold, h, cond.permute(0, 2, 1))
                assert wl[b] == 240 * sum(durs) + 64, 'e2e: durations of sentence %d differ from the oracle' % b
                dlt = np.abs(to16(wav[b, 0, :wl[b]].cpu().numpy()).astype(np.int32) - to16(ref.numpy().squeeze()).astype(np.int32))
                worst = max(worst, int(dlt.max()))
            assert worst <= 4, 'e2e: %d LSB from the oracle chain' % worst
            efl, egen = e2e_flops(float(sum(wl)) / 240.0, float(sum(int(v) for v in lens[:xx.shape[0]])))
            legs[tag] = {'ms': dt * 1e3, 'samples': int(sum(wl)), 'samples_per_s': float(sum(wl)) / dt, 'max_lsb_vs_oracle_chain': worst,
                         'roofline': {'bound': 'mfma', 'achieved': efl / dt / 1e12, 'peak': PEAK_F16_MFMA_TFLOPS / 3, 'unit': 'TFLOP/s',
                                      'frac': efl / dt / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3), 'traffic': None, 'generator_share_of_flops': egen / efl}}
        del tts
    except Exception as e:
        legs['e2e_64_sentences'] = {'error': str(e)[:200]}
    return legs


def cpu_baseline(h, sd, mel_dev, budget_s=10.0):
    """Oracle (torch fp32 CPU restatement of the reference generator) on the host cores, on a bounded sample of the BENCHED
    inputs (the first 2 utterances, first 300 frames).  Fixed protocol: min(32, cpu_count) threads — torch's
    conv kernels stop scaling there, 256 logical CPUs oversubscribe them — plus a 1-thread figure on one 1-second utterance."""
    import torch
    from oracle import hifigan_ref as R
    ncpu = os.cpu_count() or 1
    cores = min(32, ncpu)
    w = R.fold_state_dict(sd)
    mel = mel_dev[:2, :, :300].detach().cpu().contiguous()
    B, T = mel.shape[0], mel.shape[2]
    with torch.no_grad():
        torch.set_num_threads(1)
        one = mel[:1, :, :100].contiguous()
        R.generator_forward(w, h, one[:, :, :20])
        t0 = time.perf_counter()
        o1 = R.generator_forward(w, h, one)
        t1 = time.perf_counter() - t0
        def passes(nthreads, budget):
            # 1 warm-up, then whole passes over the sample until the budget is spent (at least 5): per-pass times
            torch.set_num_threads(nthreads)
            R.generator_forward(w, h, mel[:, :, :40])
            times, t_start = [], time.perf_counter()
            while True:
                t0 = time.perf_counter()
                out = R.generator_forward(w, h, mel)
                times.append(time.perf_counter() - t0)
                if (time.perf_counter() - t_start > budget and len(times) >= 5) or len(times) >= 200:
                    break
            return times, out
        times, out = passes(cores, budget_s)
        el = sum(times)
        n = len(times)
        med = sorted(times)[n // 2]
        # BASELINE.md §4 also asks for os.cpu_count() threads: measured beside the fixed-32 figure on ONE 40-frame utterance — on this pool's hosts (256
        # logical CPUs) torch's convolution kernels collapse when oversubscribed (2.6 k samples/s measured: 300 x slower than on 32 threads), and the
        # default bench line must finish within minutes
        if ncpu != cores:
            torch.set_num_threads(ncpu)
            tiny = mel[:1, :, :40].contiguous()
            R.generator_forward(w, h, tiny[:, :, :10])
            t0 = time.perf_counter()
            oa = R.generator_forward(w, h, tiny)
            all_rate = oa.shape[2] / (time.perf_counter() - t0)
            torch.set_num_threads(cores)
        else:
            all_rate = out.shape[0] * out.shape[2] / med
    per_pass = out.shape[0] * out.shape[2]
    return {'value': per_pass / med, 'unit': 'audio samples/s', 'cores': cores, 'kind': 'port',
            'value_1_thread': o1.shape[2] / t1, 'value_mean': per_pass * n / el,
            'value_cpu_count_threads': all_rate, 'cpu_count': ncpu,
            'sample': 'median of %d passes (1 warm-up) of oracle generator_forward over the first %d utterances x %d frames of the benched batch, %.1f s in all, '
                      'torch CPU fp32, %d threads (host has %d logical CPUs; value_cpu_count_threads = one 40-frame utterance with torch.set_num_threads(%d)); '
                      '1-thread figure: one 100-frame utterance in %.1f s'
                      % (n, B, T, el, cores, ncpu, ncpu, t1)}


def bench_train(args):
    """`--mode train`: BASELINE configs[3] per-GPU share — the full Cubegan adversarial training step (discriminator step,
    generator step, text step; cube/networks/cubegan.py:85-189) on synthetic examples, b utterances x 12 000-sample crops per
    GPU, data parallel: replicated parameters, rank-distinct data, three flat-bucket RCCL exchanges per step
    (reduce_scatter + all_gather over xGMI, ttscube_amd/distributed.py).  The exchange runs at N = 1 too (a world of one), so the
    single-GPU line times the same code path.  Weak scaling: the per-GPU batch is fixed."""
    import random
    import torch
    import torch.distributed as dist
    from ttscube_amd.distributed import broadcast_parameters
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate
    from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
    from ttscube_amd.networks import training as T
    from ttscube_amd.networks.cubegan import Cubegan

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, '--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29517')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    if getattr(args, 'miopen_find', False):
        torch.backends.cudnn.benchmark = True   # the discriminators' strided / grouped convolutions still run on MIOpen
    b = args.train_batch
    enc = synthetic_encodings()
    torch.manual_seed(1234 + rank)                       # ranks start different on purpose: broadcast must make them equal
    model = Cubegan(enc, conditioning=None, train=True).to(dev)
    model.train()
    broadcast_parameters(model)
    opts = T.cubegan_configure_optimizers(model)
    reducers = T.cubegan_reducers(model, opts, force=True)   # the exchange runs at N = 1 too (a world of one)
    batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(b, 777 + rank, min_ph=30, max_ph=50)))   # rank-distinct data
    crop = random.Random(99 + rank)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        out = T.cubegan_training_step(model, batch, opts, reducers, rng=crop)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = T.cubegan_training_step(model, batch, opts, reducers, rng=crop)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    # replicas must still be identical after the timed steps
    chk = torch.stack([p.detach().double().abs().sum() for p in model.parameters()]).sum().reshape(1)
    allchk = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allchk, chk)
    same = all(bool(torch.equal(c, allchk[0])) for c in allchk)
    assert same, 'replicas diverged: parameter checksums %s' % [float(c) for c in allchk]
    # the exchange alone, timed on its own (bytes / time = algorithm bandwidth of reduce_scatter + all_gather)
    barrier()
    t1 = time.perf_counter()
    for _ in range(5):
        for r in reducers:
            r.reduce()
    barrier()
    ex_ms = (time.perf_counter() - t1) / 5 * 1e3
    ex_bytes = sum(r.bytes_exchanged for r in reducers)
    early = [getattr(r, 'launched_early', None) for r in reducers]
    # exposed exchange time: the same steps without any exchange (only meaningful with one rank: replicas would diverge otherwise)
    exposed_ms = None
    if world == 1:
        for _ in range(2):
            T.cubegan_training_step(model, batch, opts, None, rng=crop)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            T.cubegan_training_step(model, batch, opts, None, rng=crop)
        torch.cuda.synchronize()
        exposed_ms = elapsed / args.steps * 1e3 - (time.perf_counter() - t2) / args.steps * 1e3
    if rank == 0:
        samples = world * b * 12000
        nparam = sum(p.numel() for p in model.parameters())
        res = {'metric': 'audio samples/sec (HiFi-GAN adversarial training step, Cubegan)', 'value': samples * args.steps / elapsed,
               'unit': 'samples/s', 'n_gpus': world, 'rccl_ranks': (dist.get_world_size() if dist.is_initialized() else 0), 'steps': args.steps, 'warmup': max(1, args.warmup),
               'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f32 (convolutions: split fp16 hi/lo x3 MFMA with per-launch device-side ranges; TTSC_TRAIN_SPLIT=0 = exact fp32 MFMA)', 'data': 'synthetic',
               'config': {'workload': 'Cubegan.training_step (D + G + text steps, 4 optimizers), %d utterances x 12000-sample crops per GPU, '
                                      'generator + Languasito2 + MPD + MSD = %.1f M parameters' % (b, nparam / 1e6),
                          'global_batch': world * b, 'parallelism': 'dp%d: replicated parameters, 3 flat-bucket RCCL exchanges per step '
                                                                    '(reduce_scatter + all_gather)' % world},
               'exchange': {'bytes_per_step_per_rank': ex_bytes, 'ms_per_step_alone': ex_ms,
                            'algorithm_GBs': ex_bytes / (ex_ms * 1e-3) / 1e9 if ex_ms > 0 else None, 'replicas_identical': same,
                            'exposed_ms_per_step': exposed_ms, 'chunks_launched_during_backward': early, 'note': 'reduce_scatters leave from bucket-ready gradient hooks during backward(); '
                            'exposed = step time with the three exchanges minus step time without them (N = 1 only)'},
               'losses': {k: round(float(v), 5) for k, v in out.items()},
               'roofline': None, 'cpu_baseline': None}
        fl, parts = cubegan_step_conv_flops(world * b)
        ach = fl * args.steps / elapsed / 1e12 / world          # per-GPU TFLOP/s
        res['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': 2500.0 / 3, 'unit': 'TFLOP/s', 'frac': ach / (2500.0 / 3), 'traffic': None,
                           'kernel': 'conv_f16x3 forward / data-gradient launches + wgrad_f16x3_kernel (split fp16 hi/lo x3 MFMA) of the generator and the '
                                     'MPD / MSD discriminators: all convolution launches of one step',
                           'flops_per_step': fl, 'flops_model': 'bench.py::cubegan_step_conv_flops (3 x generator forward + 9 x discriminator forward over b sequences)',
                           'note': 'whole-step time in the denominator (LSTM recurrences, losses, optimizers, exchange included): a lower bound on the convolution kernels\' own fraction'}
        print(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


def cubegan_step_conv_flops(b, crop_frames=50):
    """Algorithmic convolution FLOPs of ONE Cubegan.training_step at `b` crops of `crop_frames` frames (12 000 samples) — the numerator
    of the `--mode train` roofline.  Generator (HiFi-GAN V1, SURVEY §8d: 584 279 MAC per output sample): forward + data gradient +
    weight gradient = 3 x forward.  Discriminators (MPD periods 2/3/5/7/11, MSD 3 scales; layer tables of hifigan/discriminators.py,
    grouped layers counted with Cin / groups): the D step runs them forward on real + generated (2b sequences) and differentiates
    everything (3 x forward); the G step runs them forward on 2b again and takes the data gradient of the generated half only
    (2 + 1 forward units of b).  The LSTM / Linear GEMMs of the text stacks (~1 % of this) are not counted."""
    L = crop_frames * 240
    gen = 584279.4 * 2.0 * b * (L + 64)
    mac = 0.0
    for p in (2, 3, 5, 7, 11):                      # DiscriminatorP on one sequence
        Hh = -(-L // p)
        ch = [1, 32, 128, 512, 1024]
        for i in range(4):
            Hh = (Hh + 2 * 2 - 5) // 3 + 1
            mac += Hh * p * ch[i] * ch[i + 1] * 5
        mac += Hh * p * 1024 * 1024 * 5 + Hh * p * 1024 * 3
    Ls = L
    for s in range(3):                              # DiscriminatorS on one sequence, scale s
        if s:
            Ls = (Ls + 2 * 2 - 4) // 2 + 1
        Lc = Ls
        for cin, cout, k, st, g in ((1, 128, 15, 1, 1), (128, 128, 41, 2, 4), (128, 256, 41, 2, 16), (256, 512, 41, 4, 16),
                                    (512, 1024, 41, 4, 16), (1024, 1024, 41, 1, 16), (1024, 1024, 5, 1, 1), (1024, 1, 3, 1, 1)):
            Lc = (Lc + 2 * ((k - 1) // 2) - k) // st + 1
            mac += Lc * cout * (cin // g) * k
    disc_fwd_b = 2.0 * mac * b                      # FLOPs of one forward over b sequences
    total = 3.0 * gen + (3.0 * 2 + 2 + 1) * disc_fwd_b
    return total, {'generator_fwd': gen, 'discriminators_fwd_per_b_sequences': disc_fwd_b}


def e2e_flops(frames_total, phones_total):
    """Algorithmic FLOPs of synthesize() over a set of sentences: the generator's 584 279 MAC per output sample (240 samples per
    frame + 64 per sentence ignored) and Languasito2's per-frame / per-phoneme GEMM + recurrence work (~3.9 M MAC per
    frame for the pitch + conditioning BiLSTMs; per phoneme: two char stacks 64->256 k3 x3 convs + 2 BiLSTM(256) layers each + the
    duration BiLSTM)."""
    gen = 2.0 * 584279.4 * 240.0 * frames_total
    per_frame = 2.0 * 3.9e6 * frames_total
    lstm = lambda i, hh: 2 * 4 * hh * (i + hh)
    per_phone_mac = 2 * (64 * 256 * 3 + 2 * 256 * 256 * 3 + lstm(256, 256) + lstm(512, 256)) + lstm(640, 256) + lstm(512, 256)
    return gen + per_frame + 2.0 * per_phone_mac * phones_total, gen


def bench_e2e(args):
    """`--mode e2e`: BASELINE configs[4] — full synthesize(): phoneme ids -> BiLSTM mel decoder (Languasito2) -> HiFi-GAN, random
    sentences (20..120 phonemes) sharded over the GPUs of one node, 64 sentences per GPU (N = 8: the 512 sentences of the config),
    no collective on the data path (`TTSCube.shard`, cube/api.py:45-66 is the B = 1 loop this replaces).  A step = one pass over this
    rank's shard as length-bucketed padded batches.  Per-phase device time (text stacks / alignment / frame stacks / generator) from
    HIP events; never a time for wrong audio: the shortest and the longest sentence of every rank are checked against the oracle
    chain (meldecoder_ref -> hifigan_ref: identical durations, <= 4 LSB int16)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import hifigan_ref as R          # synthetic weights + the post-run check only
    from oracle import meldecoder_ref as MO
    from ttscube_amd.api import TTSCube
    from ttscube_amd.io_utils.synthetic import synthetic_sentences
    from ttscube_amd.networks.cubegan import Cubegan

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, '--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    class _Enc:
        phon2int = {'p%d' % i: i for i in range(50)}
        speaker2int = {'s0': 0}
        max_pitch = 300
        max_duration = 12   # synthetic weights give ~uniform durations: ~6 frames per phoneme

    h = dict(R.CONFIG_V1)
    gsd = R.synthetic_state_dict(h, seed=1234)
    torch.manual_seed(0)
    tts = Cubegan(_Enc(), conditioning=None, train=False)
    sd = tts.state_dict()
    lsd = MO.fill_state_dict(MO.named_shapes(tts._languasito), 5)
    sd.update({'_languasito.' + k: v for k, v in lsd.items()})
    sd.update({'_generator.' + k: v for k, v in gsd.items()})
    tts.load_state_dict(sd)
    tts = tts.to(dev).eval()
    per_gpu, bs = args.sentences_per_gpu, args.e2e_batch
    xc, lens = synthetic_sentences(per_gpu * world, seed=1234)
    mine = TTSCube.shard(list(range(per_gpu * world)), rank, world)
    order = sorted(mine, key=lambda i: int(lens[i]))                  # length buckets: neighbours in length share a padded batch
    batches = [order[s:s + bs] for s in range(0, len(order), bs)]

    def run(timers=None):
        outs, n = {}, 0
        for b in batches:
            L = int(max(lens[i] for i in b))
            X = {'x_char': torch.from_numpy(xc[b][:, :L]), 'x_speaker': torch.ones((len(b), 1), dtype=torch.long)}
            # check='deferred': the generator's range guard does not stall the pipeline (the host prepares the next batch's text stack
            # while this batch's generator runs); every batch's verdict is still collected inside the timed pass
            wav, wl = tts.inference(X, return_lengths=True, timers=timers, check='deferred')
            n += int(sum(wl))
            outs.update({i: (wav[k, 0, :wl[k]], wl[k]) for k, i in enumerate(b)})
        tts._generator.finish_range_check()
        return outs, n

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_pipelined(nsteps):
        """`nsteps` passes over this rank's shard as ONE stream of batches through Cubegan.inference_pipelined: the text / frame stacks of
        the next batch run under the generator of the current one (two streams); returns the last pass's outputs"""
        import itertools
        def feed():
            for _ in range(nsteps):
                for b in batches:
                    L = int(max(lens[i] for i in b))
                    yield {'x_char': torch.from_numpy(xc[b][:, :L]), 'x_speaker': torch.ones((len(b), 1), dtype=torch.long)}
        outs, n = {}, 0
        for k, (wav, wl) in enumerate(tts.inference_pipelined(feed(), check='deferred', lstm_group=args.lstm_group)):
            if k >= (nsteps - 1) * len(batches):
                b = batches[k % len(batches)]
                n += int(sum(wl))
                outs.update({i: (wav[j, 0, :wl[j]], wl[j]) for j, i in enumerate(b)})
        return outs, n

    for _ in range(max(1, args.warmup)):
        outs, nsamp = run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs, nsamp = run()
    barrier()
    elapsed_seq = time.perf_counter() - t0
    elapsed = elapsed_seq
    pipelined = not getattr(args, 'no_pipeline', False)
    if pipelined:
        run_pipelined(max(1, args.warmup))
        barrier()
        t0 = time.perf_counter()
        outs_p, nsamp_p = run_pipelined(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        assert nsamp_p == nsamp and all(bool(torch.equal(outs_p[i][0], outs[i][0])) for i in outs), 'pipelined synthesis differs from the sequential one'
        outs = outs_p
    # per-phase device time of one more (untimed) pass
    tm = []
    run(tm)
    torch.cuda.synchronize()
    phases = {}
    for (n0, e0), (n1, e1) in zip(tm[:-1], tm[1:]):
        if n1 != 'start':
            phases[n1] = phases.get(n1, 0.0) + e0.elapsed_time(e1)
    # oracle check: shortest + longest sentence of this rank
    to16 = lambda a: np.asarray(a * 32767, dtype=np.int16)
    wfold = R.fold_state_dict(gsd)
    worst = 0
    for i in sorted({order[0], order[-1]}):
        with torch.no_grad():
            cond, durs, _ = MO.languasito2_inference(lsd, torch.from_numpy(xc[i:i + 1, :lens[i]]), torch.tensor([[1]]), _Enc.max_pitch)
            ref = R.generator_forward(wfold, h, cond.permute(0, 2, 1))
        w, wl = outs[i]
        assert wl == 240 * sum(durs) + 64, 'e2e: durations of sentence %d differ from the oracle' % i
        worst = max(worst, int(np.abs(to16(w.cpu().numpy()).astype(np.int32) - to16(ref.numpy().squeeze()).astype(np.int32)).max()))
    assert worst <= 4, 'e2e: %d LSB from the oracle chain on rank %d' % (worst, rank)
    agg = torch.tensor([elapsed, float(nsamp), float(worst)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = agg.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, nsamp_all, worst = float(mx[0]), float(sm[1]), int(mx[2])
    else:
        nsamp_all = float(nsamp)
    if rank == 0:
        value = nsamp_all * args.steps / elapsed
        print(json.dumps({
            'metric': 'audio samples/sec (end-to-end synthesize(): phonemes -> BiLSTM mel decoder -> HiFi-GAN)', 'value': value, 'unit': 'samples/s',
            'n_gpus': world, 'rccl_ranks': (dist.get_world_size() if dist.is_initialized() else 0), 'steps': args.steps, 'warmup': max(1, args.warmup), 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 (generator: split fp16 hi/lo x3 MFMA)', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[4]: %d random sentences per GPU (20-120 phonemes, %d in all), Languasito2 + HiFi-GAN V1, '
                                   'length-bucketed padded batches of %d' % (per_gpu, per_gpu * world, bs),
                       'global_batch': per_gpu * world, 'parallelism': 'utterance shards (TTSCube.shard), no collective'},
            'rtf_24k': value / world / 24000.0, 'sentences_per_s': per_gpu * world * args.steps / elapsed,
            'phase_ms_rank0': {k: round(v, 3) for k, v in phases.items()}, 'samples_per_step_rank0': int(nsamp),
            'pipelined': pipelined, 'ms_per_step_sequential_rank0': elapsed_seq / args.steps * 1e3,
            'pipeline_note': 'value / ms_per_step: all steps as one stream of batches through Cubegan.inference_pipelined (text + frame stacks of batch k+1 on one stream under the generator of batch k on another; outputs bit-identical to the sequential pass, asserted); ms_per_step_sequential = one batch after the other on one stream; phase_ms from a sequential pass',
            'max_lsb_vs_oracle_chain': worst, 'oracle_checked': '2 sentences per rank (shortest, longest)',
            'roofline': (lambda fl_gen: {'bound': 'mfma', 'achieved': fl_gen[0] * args.steps / elapsed / 1e12, 'peak': 2500.0 / 3, 'unit': 'TFLOP/s',
                                         'frac': fl_gen[0] * args.steps / elapsed / 1e12 / (2500.0 / 3), 'traffic': None,
                                         'kernel': 'generator launches (split fp16 hi/lo x3 MFMA) + Languasito2 GEMMs / recurrences (fp32): whole synthesize() pass of rank 0',
                                         'flops_per_step': fl_gen[0], 'generator_share_of_flops': fl_gen[1] / fl_gen[0],
                                         'generator_phase_frac': (fl_gen[1] / (phases['generator'] * 1e-3) / 1e12 / (2500.0 / 3)) if phases.get('generator') else None,
                                         'note': 'rank 0\'s shard over rank 0\'s wall time; generator_phase_frac = generator FLOPs over the generator phase\'s device time'})(
                e2e_flops(float(nsamp) / 240.0, float(sum(int(lens[i]) for i in mine)))),
            'cpu_baseline': None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=('infer', 'train', 'e2e'), default='infer',
                    help="'train': the Cubegan adversarial training step (BASELINE configs[3]); 'e2e': text -> audio, sentences sharded over the GPUs (configs[4])")
    ap.add_argument('--sentences-per-gpu', type=int, default=64, help='--mode e2e: sentences in every rank\'s shard')
    ap.add_argument('--e2e-batch', type=int, default=64, help='--mode e2e: sentences per padded batch')
    ap.add_argument('--lstm-group', type=int, default=4, help='--mode e2e, pipelined pass: utterances per member group of the split LSTM recurrences (ttsc_lstm_set_group_size)')
    ap.add_argument('--no-pipeline', action='store_true', help='--mode e2e: time the sequential pass only (no two-stream pipelining across batches)')
    ap.add_argument('--train-batch', type=int, default=16, help='utterances per GPU in --mode train')
    ap.add_argument('--miopen-find', action='store_true', help="--mode train: let MIOpen search its convolution algorithms exhaustively "
                    "(torch.backends.cudnn.benchmark): ~12 minutes once per process on a fresh box, then 108 instead of 145 ms per step; "
                    "scripts/train_cubegan.py --miopen-find does the same for real training runs")
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=800, help='mel frames per utterance (800 = 8 s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary legs (fp32, B=1, WaveRNN)')
    ap.add_argument('--no-live-traffic', action='store_true', help='keep roofline.traffic from the committed PMC summary instead of measuring it with two rocprofv3 --pmc child passes (N = 1, default run only)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run, rendezvous on
        # 127.0.0.1 — the container hostname may not resolve) and hand their exit code back.  Under a launcher WORLD_SIZE is set and this is skipped.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.mode == 'train':
        return bench_train(args)
    if args.mode == 'e2e':
        return bench_e2e(args)

    import torch
    import torch.distributed as dist
    from oracle import hifigan_ref as R  # synthetic weights/inputs + cpu_baseline leg only
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, '--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world)
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
    check_rms = self_check(g, mel, out, h, sd) if rank == 0 else None   # never report a time for wrong results

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
            peak, kname = PEAK_F16_MFMA_TFLOPS / 3.0, 'conv_f16x3_wide_kernel + rbchain_f16x3_kernel + conv_f16x3_kernel (v_mfma_f32_32x32x16_f16 x3 split-precision implicit-GEMM conv; all launches of one forward)'
            executed = 3.0 * achieved
        else:
            peak, kname = PEAK_FP32_MFMA_TFLOPS, 'conv_mfma_kernel (v_mfma_f32_32x32x2_f32 implicit-GEMM conv; all launches of one forward)'
            executed = achieved
        traffic = measured_traffic(precision, B, T)
        res = {
            'metric': 'audio samples/sec (HiFi-GAN vocoder inference)', 'value': value, 'unit': 'samples/s',
            'n_gpus': world, 'rccl_ranks': (dist.get_world_size() if dist.is_initialized() else 0), 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
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
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'HBM bytes per step (rocprofv3 PMC, %s)' % os.path.relpath(TRAFFIC_CSV, ROOT),
                         'traffic_source': 'committed PMC summary of this workload (builder-run rocprofv3 passes), NOT measured in this run',
                         'kernel': kname,
                         'flops_per_step': flops_step, 'device_ms_per_step': dev_ms / args.steps,
                         'mfma_executed_tflops': executed, 'mfma_dense_peak_tflops': PEAK_F16_MFMA_TFLOPS if precision == 'f16x3' else PEAK_FP32_MFMA_TFLOPS,
                         'x_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'hbm_compulsory_GBs': alg_bytes * args.steps / (dev_ms * 1e-3) / 1e9,
                         'hbm_frac_compulsory': alg_bytes * args.steps / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         # how far the fusion goes: measured HBM bytes against the model in which every convolution reads its input (+ residual) and
                         # writes its output through HBM (12.5 KB per output sample)
                         'traffic_ratio_vs_unfused': (traffic / (UNFUSED_BYTES_PER_SAMPLE * B * Lout)) if traffic else None,
                         # matrix instructions of one forward: what the arithmetic needs (3 split products per 32x32x16 tile step) and what the kernels
                         # issue (halo recomputation, channel / column padding included; committed counter summary of this workload)
                         'mfma_insts_per_forward': {'algorithmic': (3.0 if precision == 'f16x3' else 16.0) * flops_step / (2.0 * 32 * 32 * 16),
                                                    'issued_pmc': measured_mfma_insts() if precision == 'f16x3' else None,
                                                    'source': os.path.relpath(PMC_CSV, ROOT) + ' (SQ_INSTS_MFMA, builder-run rocprofv3 pass, NOT measured in this run)'}},
        }
        res['self_check_rms_vs_oracle'] = check_rms
        if precision == 'f16x3':
            # What the matrix pipe of THIS device sustains, measured now (csrc/probe.hip: register-only v_mfma_f32_32x32x16_f16 loop on every CU):
            # MI355X clocks to its power budget, so operands with real bit patterns run well below the nominal peak that `peak` must quote.
            # `frac` stays achieved / nominal; `frac_of_sustained` prices the same number against the split-mix loop measured in this run.
            try:
                import ctypes as C
                from ttscube_amd import _lib
                sus = {}
                for mode, name in ((0, 'zeros'), (1, 'random_fp16'), (2, 'split_mix')):
                    tf = C.c_double(0.0)
                    _lib.check(_lib.lib().ttsc_probe_mfma_tflops(mode, 60.0, C.byref(tf), _lib.current_stream()), 'ttsc_probe_mfma_tflops')
                    sus[name] = tf.value
                res['roofline']['sustained_f16_mfma_tflops'] = sus
                res['roofline']['frac_of_sustained'] = executed / sus['split_mix']
                res['roofline']['sustained_note'] = ('register-only MFMA loop, all CUs, ~60 ms per operand pattern, measured in this run after the timed region; '
                                                     'split_mix = the three products of the split-precision scheme on random data: the ceiling an MFMA-bound '
                                                     'split-precision kernel has on this chip at its power budget')
            except Exception as e:
                res['roofline']['sustained_f16_mfma_tflops'] = {'error': str(e)[:200]}
        if world == 1 and not args.no_extra:
            res['extra'] = extra_legs(g, h, sd, dev, R)
            if precision == 'f16x3':   # the same workload on the exact fp32 MFMA (k-ordered fmaf chain), for the record
                try:
                    g.set_precision('fp32')
                    ms32, o32 = time_forward(g, mel, max(2, args.steps // 4), 1)
                    res['extra']['exact_fp32_same_workload'] = {'ms_per_step': ms32, 'samples_per_s': B * Lout / (ms32 * 1e-3),
                                                                'algorithmic_tflops': flops_step / (ms32 * 1e-3) / 1e12,
                                                                'frac_of_fp32_mfma_peak': flops_step / (ms32 * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                                                'rms_vs_oracle': self_check(g, mel, o32, h, sd)}
                    g.set_precision('f16x3')
                except Exception as e:
                    res['extra']['exact_fp32_same_workload'] = {'error': str(e)[:200]}
        if world == 1 and not args.no_extra and not args.no_live_traffic and precision == 'f16x3':
            # roofline.traffic measured NOW (after every timed region of this process): two rocprofv3 --pmc child passes of this workload
            lt, detail = live_traffic(precision, B, T)
            if lt is not None:
                res['roofline']['traffic_committed_summary'] = res['roofline']['traffic']
                res['roofline']['traffic'] = lt
                res['roofline']['traffic_unit'] = 'HBM bytes per step (rocprofv3 PMC, measured in this run)'
                res['roofline']['traffic_source'] = ('measured in this run: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` child passes of `bench.py --steps 1 --warmup 0` '
                                                     'after the timed regions (bench.py::live_traffic)')
                res['roofline']['traffic_detail'] = detail
                res['roofline']['traffic_ratio_vs_unfused'] = lt / (UNFUSED_BYTES_PER_SAMPLE * B * Lout)
                res['roofline']['mfma_insts_per_forward']['issued_pmc'] = detail['sq_insts_mfma_per_forward']
                res['roofline']['mfma_insts_per_forward']['source'] = 'SQ_INSTS_MFMA, rocprofv3 --pmc child pass of this run (bench.py::live_traffic)'
            else:
                res['roofline']['traffic_live_error'] = detail
        if world == 1 and not args.no_cpu_baseline:   # reported baseline: rank 0 at N=1 only
            res['cpu_baseline'] = cpu_baseline(h, sd, mel)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
