#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through the Doppler matched-filter bank on MI355X.

One "step" = one pass of the hot path over one N-sample block already resident in HBM:
forward FFT -> Doppler search over D bins x M matched filters -> |.|^2 row sums -> Doppler pick,
including the 8-byte result read-back the reference blocks on (A3..A7 of SURVEY.md section 8).

  N=1 : config C2  (D=256 Doppler bins, M=8 GMSK matched filters, N=2^20, ov=2^10)
  N>1 : config C4  (256 bins per GPU, D=256*G sharded by bin; rank 0's block is broadcast over RCCL,
        one RCCL all-reduce of the per-bin scores per block, then the pick on every rank).
        Weak scaling: per-GPU work is fixed.  `value` counts the samples every rank pushed through
        its 256-bin bank, i.e. (N-ov) * G per step; `stream_msamples` is the physical IQ stream rate.

Search paths (--path): `segment` = single-pass overlap-save in registers/LDS (no HBM intermediate; bound
by the fp32 vector rate), `twopass` = length-N two-pass transforms through HBM (bound by HBM).  `auto`
(default) takes the segment path for short filters -- every shipped protocol.  The JSON line carries the
roofline of the path that ran the timed steps and, at N=1, a short measurement of the other one.

Launch for N>1 (driver):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N
                          --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.  Exits non-zero if the full-size parity spot check fails.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12        # B/s, MI355X spec (MI355X_MICROARCH.md)
VALU_FP32_PEAK = 157.3e12  # FLOP/s, MI355X fp32 vector peak (MI355X_MICROARCH.md)
PARITY_TOL = 1e-5        # north_star: correlation magnitudes within 1e-5
SETTLE_S = 0.05          # untimed work before the clock starts (profiles/r03_ramp.md: the clock settles within ~30 ms)


def b_alg(D, M, N):
    """Algorithmic bytes of one block through A3..A7 in the two-pass formulation."""
    return 16.0 * D * M * N + 8.0 * N * (1 + M) + 16.0 * N + 4.0 * D


def b_ref(D, M, N):
    """Traffic of the reference's unfused four-stage formulation, for context."""
    return 32.0 * D * M * N + 8.0 * N * (3 + M) + 4.0 * D * (1 + M)


def seg_flops(D, Q, Mu, L, V, bins_per_forward=1, filter_side=False):
    """Nominal flops of one block on the segment path, as the kernel that ran performs them: per (bin, segment, filter) a pointwise
    product, an inverse transform and the |.|^2 sums; forward transforms of the segments -- one per (bin, segment) together with
    the mixing multiply when the Doppler shift is applied to the samples (k_seg: the round 1-5 formula), ONE per segment and
    `bins_per_forward` bins, unmixed, when it sits on the filters' side (k_segf, round 6).  5 L log2 L per transform, 6 per
    complex multiply, 4 per accumulated output."""
    fft = 5.0 * L * np.log2(L)
    per_filter = float(D) * Q * Mu * (6.0 * L + fft + 4.0 * V)
    if filter_side:
        return per_filter + float(Q) * np.ceil(D / max(bins_per_forward, 1)) * fft
    return per_filter + float(D) * Q * (6.0 * L + fft)


def widen_range_rate(conf, radio, N, D):
    """widen rangeRateMax until the D shifts are distinct after rounding."""
    from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table
    rr = conf['Radios']['rangeRateMax']
    while True:
        _, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx'][radio], rr, N)
        if len(np.unique(shifts)) == len(shifts) or rr > 2.0e5:
            return rr, shifts
        rr *= 1.25


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(masks, shifts, x_blocks, N, ov, D, budget_s=34.0):
    """The oracle's Doppler search (numpy/scipy restatement) timed on this host's cores on a bounded sample of the same
    workload (C2 on >= 2 blocks): whole blocks -- all D bins -- of `x_blocks` while about `budget_s` seconds of CPU
    work allow (two C2 blocks on the 16-CPU share of a GPU box: 12.5 ... 14 s each on the round's hosts), else the leading bins of ONE block with the per-block figure
    scaled by D/nb (`blocks_timed` < 1 says so).  Two modes: all cores (scipy.fft complex64, workers = the CPU share) and one
    thread (numpy pocketfft, complex128, row at a time, 8 bins).  Reported baseline, not a target.  Returns the figures, the
    all-cores scores of every block timed (a list) and the single-thread scores of the first block's leading bins."""
    import scipy.fft as sfft
    from oracle import mfbank_oracle as orc
    from pycusdr_amd.hostcpu import cpu_share
    share = cpu_share()                                    # affinity and cgroup quota, not the host's core count
    M = masks.shape[0]
    group = max(1, share // M)                             # bins per inverse-FFT call: group * M rows, one per worker
    cores = min(share, group * M)
    Mw = masks.astype(np.complex64)

    def run(X, nb):
        t0 = time.perf_counter()
        out = np.zeros(nb)
        for j0 in range(0, nb, group):
            js = range(j0, min(nb, j0 + group))
            prod = np.concatenate([np.roll(X, -int(shifts[j]))[None, :] * Mw for j in js])
            y = sfft.ifft(prod, axis=-1, norm='forward', workers=cores)
            e = (y.real.astype(np.float64) ** 2 + y.imag.astype(np.float64) ** 2).sum(axis=-1) / orc.SCALE_2_18
            out[j0:j0 + len(js)] = e.reshape(len(js), M).sum(axis=1)
        return time.perf_counter() - t0, out
    X0 = orc.forward_fft(x_blocks[0])
    t1, _ = run(X0, group)                   # warm-up + calibration
    est_block = t1 * D / group
    nblk = int(min(len(x_blocks), max(0, budget_s // max(est_block, 1e-3))))
    scores, t = [], 0.0
    if nblk >= 1:
        for b in range(nblk):                # whole blocks: forward FFT included, as in the GPU step
            t0 = time.perf_counter()
            Xb = X0 if b == 0 else orc.forward_fft(x_blocks[b])
            tb, sb = run(Xb, D)
            t += (time.perf_counter() - t0) if b else tb
            scores.append(sb)
        nb, t_block, blocks_timed = D, t / nblk, float(nblk)
        what = f'{nblk} whole block(s) of 2^{int(np.log2(N))} samples, all {D} Doppler bins each (M={M})'
    else:
        nb = int(max(2, min(D, round(0.55 * budget_s * group / max(t1, 1e-3)))))
        t, sb = run(X0, nb)
        scores.append(sb)
        t_block, blocks_timed = t * D / nb, nb / D
        what = f'{nb} of {D} Doppler bins of one 2^{int(np.log2(N))}-sample block (M={M}); per-block time scaled by D/{nb}'
    # single thread, numpy complex128: 8 bins
    ns = min(8, D)
    t0 = time.perf_counter()
    single = orc.doppler_scores(X0, masks, shifts[:ns], True)[:, 0]
    ts = time.perf_counter() - t0
    return {
        'value': round((N - ov) / t_block / 1e6, 5), 'unit': 'Msamples/s', 'cores': cores, 'kind': 'port',
        'cpu': cpu_model(), 'blocks_timed': round(blocks_timed, 4),
        'sample': f'{what}, {t:.1f} s of scipy.fft complex64 work, {group} bins x {M} filters per call, workers={cores} '
                  f'(CPU share of this process {share} of {os.cpu_count()} host cores)',
        'single_thread': {'value': round((N - ov) / (ts * D / ns) / 1e6, 6), 'unit': 'Msamples/s', 'cores': 1,
                          'sample': f'{ns} of {D} bins, numpy.fft complex128, {ts:.1f} s; scaled by D/{ns}'},
    }, scores, single


def kfd_gpu_count():
    """GPUs of this node as the kernel driver lists them (/sys/class/kfd/kfd/topology/nodes: a node with SIMDs is a GPU), read
    WITHOUT touching the HIP runtime -- the parent of the rank processes must never initialise the GPU.  None when the
    topology is not readable."""
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(base):
            for line in open(os.path.join(base, node, 'properties')):
                k, _, v = line.partition(' ')
                if k == 'simd_count' and int(v) > 0:
                    n += 1
        return n
    except (OSError, ValueError):
        return None


def chain_figures(local_rank, blocks_per_call=None, n_packets=240):
    """The whole receive chain at the reference's own block geometry (config/base.json:13,33: blocks of 2^15 ... 2^17 samples, 64
    bins): host chunks of 2^14 samples in (examples/benchmark/bench_modem.py:32), page-locked window, H2D, A3 ... A13, result
    dicts out -- `recv_*`, the loop the reference's Demodulator_process runs (DP:284-338) -- and the same with the decoder in the
    same thread, packets out -- `chain_*` (in the reference the decoder is another process).  B consecutive blocks per device
    call (mfb_receive_blocks_*); `*_b1_*` is the one-block loop."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_modem', os.path.join(ROOT, 'examples', 'benchmark', 'bench_modem.py'))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    out = {}
    per_call = blocks_per_call or {15: 32, 17: 8}             # windows of 2^20 samples (DemodulatorRunner.auto_blocks_per_call)
    for log2n in (15, 17):
        Bn = per_call[log2n] if isinstance(per_call, dict) else per_call
        stim = bm.make_stream('GMSK', n_packets, 12.0, log2n, 2)          # one stimulus for every run at this size
        bm.run_snr('GMSK', n_packets, 12.0, log2n, 'transforms', 2, 64, blocks_per_call=Bn, stimulus=stim)    # handles, code objects, clock
        out[f'chain_n{log2n}_blocks_per_call'] = Bn
        # '' = B configured; '_auto' = nothing configured, a plain chunk iterator: the loop batches what the source has ready
        # (run_stream, adaptive); '_b1' = "blocks_per_call": 1, the reference's one-block loop.  The loop is host-bound and shares
        # its CPUs with whatever else runs on the box: three runs per figure, the configured and the adaptive loop taking turns
        # (a drifting host hits both alike), the median of each
        runs = {}
        for rep in range(3):
            for B, tag in ((Bn, ''), ('auto', '_auto')) + (((1, '_b1'),) if rep == 0 else ()):
                for decode, name in ((True, 'chain'), (False, 'recv')):
                    runs.setdefault((name, tag, decode), []).append(
                        bm.run_snr('GMSK', n_packets, 12.0, log2n, 'transforms', 2, 64, blocks_per_call=B, decode=decode, stimulus=stim))
        for (name, tag, decode), rr in runs.items():
            r = sorted(rr, key=lambda q: q['ksamples_per_s'])[len(rr) // 2]
            out[f'{name}{tag}_n{log2n}_d64_msamples'] = round(r['ksamples_per_s'] / 1e3, 1)
            if tag != '_b1':
                out[f'{name}{tag}_n{log2n}_d64_best'] = round(max(q['ksamples_per_s'] for q in rr) / 1e3, 1)
            if decode:
                out[f'{name}{tag}_n{log2n}_d64_packets'] = f"{r['packets']}/{r['sent']}"
    return out


def c5_concurrent_figures(seconds=2.0, D=512):
    """BASELINE C5 as worded: two CONCURRENT demodulator instances -- CC11xx (FSK-2 at 128 samples per symbol, 384 taps) and the
    custom BPSK filter set (M = 32) -- at 512 bins, N = 2^20, on ONE MI355X, each a process and device context of its own (as
    the reference runs its radios: pyCuSDR.py:245-251, DB:177-181).  Fresh children (tools/c5_rate_child.py), first one after the
    other, then started together on the same moment; every leg `seconds` of the search step on resident blocks."""
    import subprocess
    child = os.path.join(ROOT, 'tools', 'c5_rate_child.py')

    def run(names, shared=False):
        procs = [subprocess.Popen([sys.executable, child, n, str(D), str(seconds)] + ([str(i), str(len(names))] if shared else []),
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
                 for i, n in enumerate(names)]
        import select

        def line_of(p, timeout_s):
            # (a child that hangs must cost this leg its figures, not the bench its line)
            if not select.select([p.stdout], [], [], timeout_s)[0]:
                raise RuntimeError(f'c5 child silent for {timeout_s} s')
            return p.stdout.readline()
        try:
            for p in procs:
                line = line_of(p, 90.0)
                if line.strip() != 'ready':
                    raise RuntimeError(f'c5 child said {line!r}')
            t_go = time.time() + 0.4
            for p in procs:
                p.stdin.write(f'go {t_go}\n')
                p.stdin.flush()
            outs = [json.loads(line_of(p, 30.0 + seconds)) for p in procs]
            for p in procs:
                p.wait(timeout=60)
            return outs
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    out = {}
    try:
        (cc,), (bp,) = run(['CC11xx']), run(['bench_BPSK'])
        cc2, bp2 = run(['CC11xx', 'bench_BPSK'])
        out = {'c5_cc11xx_alone': cc['msamples'], 'c5_bpsk_alone': bp['msamples'], 'c5_cc11xx_beside': cc2['msamples'],
               'c5_bpsk_beside': bp2['msamples'],
               'c5_sum_over_alone': round(cc2['msamples'] / cc['msamples'] + bp2['msamples'] / bp['msamples'], 4),
               'c5_seconds_per_leg': seconds, 'c5_bins': D,
               'c5_note': 'two processes, two device contexts, one MI355X, started on the same moment; 1.0 = the device is shared '
                          'without loss, < 1 = what time-slicing between the two contexts costs'}
    except Exception as e:       # noqa: BLE001 -- a secondary figure must not cost the line
        out = {'c5_error': str(e)[:110]}
    try:
        # ... and with the device's compute units dealt out between the two (mfb_set_cu_share: even CUs / odd CUs)
        cc3, bp3 = run(['CC11xx', 'bench_BPSK'], shared=True)
        out.update({'c5_cc11xx_shared': cc3['msamples'], 'c5_bpsk_shared': bp3['msamples']})
        if 'c5_cc11xx_alone' in out:
            out['c5_shared_sum_over_alone'] = round(cc3['msamples'] / out['c5_cc11xx_alone'] + bp3['msamples'] / out['c5_bpsk_alone'], 4)
    except Exception as e:       # noqa: BLE001
        out['c5_shared_error'] = str(e)[:110]
    return out


def c5_inprocess_figures(dev, local_rank, blocks, esz, nblocks, seconds=1.0, D=512, log2N=20):
    """The same two instances as two handles of ONE process (two streams, two threads; the library calls release the
    interpreter lock): alone, then beside each other."""
    import threading
    import torch
    from pycusdr_amd import config as cfg
    from pycusdr_amd.mfbank import MFBank
    from pycusdr_amd.protocol import loadProtocol
    N, ov = 1 << log2N, 1 << 10
    banks = {}
    try:
        for name in ('CC11xx', 'bench_BPSK'):
            if name == 'CC11xx':
                conf, sps, msz = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D, device=local_rank), 128, 3
            else:
                conf, sps, msz = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D, device=local_rank), 16, 5
            rr, shifts = widen_range_rate(conf, 'UHF-H', N, D)
            conf['Radios']['rangeRateMax'] = rr
            M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, msz)
            b = MFBank(log2N, D, M, window_width=7, sum_all_masks=True, device=local_rank)
            b.set_filters(masks)
            b.set_shifts(shifts)
            banks[name] = b

        def loop(b, res, key, t_end):
            i = n = 0
            while time.perf_counter() < t_end[0] - seconds:         # settle until the common start
                b.upload_device(blocks.data_ptr() + (i % nblocks) * esz)
                b.find_carrier()
                i += 1
            t0 = time.perf_counter()
            while time.perf_counter() < t_end[0]:
                b.upload_device(blocks.data_ptr() + (i % nblocks) * esz)
                b.find_carrier()
                i += 1
                n += 1
            res[key] = round((N - ov) * n / (time.perf_counter() - t0) / 1e6, 2)

        def run(names):
            res, t_end = {}, [time.perf_counter() + 0.1 + seconds]
            ths = [threading.Thread(target=loop, args=(banks[k], res, k, t_end)) for k in names]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize(dev)
            return res
        a1, a2, both = run(['CC11xx']), run(['bench_BPSK']), run(['CC11xx', 'bench_BPSK'])
        return {'c5_inproc_cc11xx_alone': a1['CC11xx'], 'c5_inproc_bpsk_alone': a2['bench_BPSK'],
                'c5_inproc_cc11xx_beside': both['CC11xx'], 'c5_inproc_bpsk_beside': both['bench_BPSK'],
                'c5_inproc_sum_over_alone': round(both['CC11xx'] / a1['CC11xx'] + both['bench_BPSK'] / a2['bench_BPSK'], 4)}
    except Exception as e:       # noqa: BLE001
        return {'c5_inproc_error': str(e)[:110]}
    finally:
        for b in banks.values():
            b.close()


def segment_roofline_core(info, Dl, Mu, launches, kernel_ms_total, sinfo=None):
    """fp32-vector roofline of the segment kernel from its own launch durations (HIP events on the library's stream).  `sinfo`:
    MFBank.get_search_info() -- which form of the kernel ran."""
    L, V, Q = 1 << info['log2L'], info['valid_per_segment'], info['segments']
    sinfo = sinfo or {'filter_side': False, 'bins_per_forward': 1}
    fl = seg_flops(Dl, Q, Mu, L, V, sinfo['bins_per_forward'], sinfo['filter_side'])
    fl_r05 = seg_flops(Dl, Q, Mu, L, V)
    launches = max(launches, 1)
    k_avg_s = kernel_ms_total / launches * 1e-3
    kname = 'k_segf' if sinfo['filter_side'] else 'k_seg'
    return {'bound': 'valu_fp32', 'kernel': f'{kname}<{L},REDUCE> ({V} valid outputs of {L}, {Q} segments per bin)',
            'achieved': round(fl / k_avg_s / 1e12, 2), 'peak': VALU_FP32_PEAK / 1e12, 'unit': 'TFLOP/s',
            'frac': round(fl / k_avg_s / VALU_FP32_PEAK, 4), 'launches': launches, 'avg_launch_ms': round(k_avg_s * 1e3, 4),
            'flops_per_launch': fl, 'filter_side_shift': bool(sinfo['filter_side']), 'bins_per_forward': int(sinfo['bins_per_forward']),
            'frac_r05_formula': round(fl_r05 / k_avg_s / VALU_FP32_PEAK, 4), 'flops_per_launch_r05_formula': fl_r05}


def twopass_roofline_core(N, Dl, Mu, counts, kms, nsteps):
    """HBM roofline of the two-pass path's dominant kernel from its own launch durations: algorithmic bytes per launch (DESIGN.md
    4.2) over the HIP-event average."""
    dom = 0 if kms[0] >= kms[1] else 1
    launches = max(counts[dom], 1)
    bins_per_launch = Dl * nsteps / launches
    if dom == 0:
        k_bytes = 8.0 * bins_per_launch * Mu * N + 8.0 * N * (1 + Mu)   # Z write + spectrum + filter bank read once
    else:
        k_bytes = 8.0 * bins_per_launch * Mu * N + 4.0 * bins_per_launch * Mu  # Z read + partial sums
    k_avg_s = kms[dom] / launches * 1e-3
    return {'bound': 'hbm', 'kernel': ['k_pass1<BANK>', 'k_pass2<REDUCE>'][dom], 'achieved': round(k_bytes / k_avg_s / 1e9, 2),
            'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': round(k_bytes / k_avg_s / HBM_PEAK, 4), 'launches': launches,
            'avg_launch_ms': round(k_avg_s * 1e3, 4), 'alg_bytes_per_launch': k_bytes}


def bank_figure(dev, local_rank, protocol, D, log2N, blocks, esz, nblocks, steps=10, warmup=2, s2_blocks=None, span=False, twopass_steps=0):
    """Untimed-region figure of another BASELINE bank on the same device: its own handle, the same step as the headline
    (forward FFT, search over D bins, pick, 8-byte read-back), HIP-event kernel time, the same flop formula.  With `s2_blocks`
    (white noise resident in HBM) the same loop once more on them, behind a settle of its own: `s2_msamples`."""
    import torch
    from pycusdr_amd import config as cfg
    from pycusdr_amd.mfbank import MFBank
    from pycusdr_amd.protocol import loadProtocol
    N, ov = 1 << log2N, 1 << 10
    if protocol == 'CC11xx':
        conf, sps, msz = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D, device=local_rank), 128, 3
    else:
        conf, sps, msz = cfg.bench_config(protocol, blockSize=log2N, doppCarrierSteps=D, device=local_rank), 16, (5 if protocol == 'bench_BPSK' else 3)
    rr, shifts = widen_range_rate(conf, 'UHF-H', N, D)
    conf['Radios']['rangeRateMax'] = rr
    proto = loadProtocol(protocol)(conf=conf)
    t0 = time.perf_counter()
    M, masks = proto.get_filter(N, sps, msz)
    t_gen = time.perf_counter() - t0
    bank = MFBank(log2N, D, M, window_width=7, sum_all_masks=True, device=local_rank)
    try:
        t0 = time.perf_counter()
        bank.set_filters(masks)
        t_set = time.perf_counter() - t0
        bank.set_shifts(shifts)
        info = bank.get_search_path()
        sinfo = bank.get_search_info()
        Mu = bank.get_info()[2]

        def leg(src, steps=steps):
            def one(i):
                bank.upload_device(src.data_ptr() + (i % nblocks) * esz)
                return bank.find_carrier()
            # untimed: at least `warmup` steps and at least 50 ms of them -- building the handle (filter generation and analysis
            # on the host) left the device idle, and it needs ~30 ms of work to settle its clock again (tools/ramp_probe.py)
            t_w, i = time.perf_counter(), 0
            while i < 1 + warmup or time.perf_counter() - t_w < SETTLE_S:
                one(i)
                i += 1
            torch.cuda.synchronize(dev)
            bank.profile_enable(True)
            t0 = time.perf_counter()
            for k in range(steps):
                one(k)
            torch.cuda.synchronize(dev)
            dt_ = (time.perf_counter() - t0) / steps
            counts_, kms_ = bank.profile_read()
            bank.profile_enable(False)
            return i, dt_, counts_, kms_
        settle, dt, counts, kms = leg(blocks)
        out = {'protocol': protocol, 'D': D, 'M': M, 'M_unique': Mu, 'samplesPerSym': sps, 'taps': info['taps'], 'path': info,
               'steps': steps, 'untimed_steps_before': settle, 'ms_per_step': round(dt * 1e3, 4), 'msamples': round((N - ov) / dt / 1e6, 2),
               'filter_generation_s': round(t_gen, 2), 'mfb_set_filters_s': round(t_set, 2),
               'rangeRateMax_used': rr, 'signal': 'S1 blocks of the headline (throughput only: the stimulus does not match this bank)'}
        if info['path'] == 'segment':
            out['roofline'] = segment_roofline_core(info, D, Mu, counts[0], kms[0], sinfo)
        if span and info['path'] == 'segment':
            # opt-in span basis of the SUM_ALL search: rank(bank) filters transformed instead of M; same table to
            # fp32 rounding -- never part of `msamples`
            try:
                sc_f = bank.get_scores()[:, 0].astype(np.float64)
                last_i = (steps - 1) % nblocks
                bank.set_search_basis('span')
                sb = bank.get_search_basis()
                _, dts, _, _ = leg(blocks)
                sc_s = bank.get_scores()[:, 0].astype(np.float64)
                out['span_basis'] = {'filters_transformed': sb[1], 'of': M, 'ms_per_step': round(dts * 1e3, 4),
                                     'msamples': round((N - ov) / dts / 1e6, 2),
                                     'max_rel_diff_vs_default_search': float(np.abs(sc_s - sc_f).max() / sc_f.max()), 'block': last_i}
            except (ValueError, RuntimeError) as e:
                out['span_basis'] = {'error': str(e)[:100]}
            finally:
                bank.set_search_basis('filters')
        if twopass_steps:
            # the HBM-bound formulation at this geometry (BASELINE C3: "HBM-bound stress, rocprof GB/s vs roofline")
            try:
                bank.set_search_path('twopass')
                _, dtt, ct_, kt_ = leg(blocks, twopass_steps)
                tp = twopass_roofline_core(N, D, Mu, ct_, kt_, twopass_steps)
                tp.update({'ms_per_step': round(dtt * 1e3, 4), 'msamples': round((N - ov) / dtt / 1e6, 2), 'steps': twopass_steps,
                           'tuning(chunk,mpb,rows,jsplit)': list(bank.get_tuning())})
                out['twopass'] = tp
            except (ValueError, RuntimeError) as e:
                out['twopass'] = {'error': str(e)[:100]}
            finally:
                bank.set_search_path('auto')
        if s2_blocks is not None:
            _, dt2, c2, k2 = leg(s2_blocks)
            out['s2_ms_per_step'] = round(dt2 * 1e3, 4)
            out['s2_msamples'] = round((N - ov) / dt2 / 1e6, 2)
            out['s2_over_s1'] = round(dt / dt2, 4)
            if info['path'] == 'segment':
                out['s2_roofline_frac'] = segment_roofline_core(info, D, Mu, c2[0], k2[0], sinfo)['frac']
        return out
    finally:
        bank.close()


def run_block_shard(args, dist, rank, G, local_rank, dev):
    """--shard blocks: the time-chunk sharding product path (pycusdr_amd.dist.BlockShard).  Every GPU holds the full D-bin
    bank; rank r runs the device stages (A3..A11) of blocks r, r + G, ...; rank 0 runs the sequential host stages (A12, A13)
    and the decoder (A14) in block order on what the owners hand back (one point-to-point message per block, no
    collective).  One step = G blocks, one per rank; value = samples of all blocks / wall time of the ordered chain."""
    import torch
    from pycusdr_amd import config as cfg, signals as sg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.dist import BlockShard
    from pycusdr_amd.protocol import loadProtocol
    log2N, ov = args.log2n, 1 << 10
    N = 1 << log2N
    conf = cfg.bench_config(args.protocol, blockSize=log2N, overlap=10, doppCarrierSteps=args.bins, device=local_rank)
    rr, shifts = widen_range_rate(conf, 'UHF-H', N, args.bins)
    conf['Radios']['rangeRateMax'] = rr
    proto = loadProtocol(args.protocol)(conf=conf)
    runner = DemodulatorRunner(conf, proto, 'UHF-H')
    group = dist.new_group(backend='gloo') if args.backend == 'nccl' else None     # the hand-back moves host arrays
    shard = BlockShard(group=group)
    nblocks = 16
    stream = sg.s1_stream(nblocks, N, ov, 'GMSK', 16, 153600, snr_db=10.0, seed=1)
    host_blocks = np.stack([stream[b * (N - ov): b * (N - ov) + N] for b in range(nblocks)])
    blocks = torch.from_numpy(host_blocks.view(np.float32).reshape(nblocks, 2 * N)).to(dev)
    esz = blocks.element_size() * 2 * N
    torch.cuda.synchronize(dev)
    decoder = Decoder(conf, proto) if rank == 0 else None

    def feed(i):
        return runner.feed_resident(blocks.data_ptr() + (i % nblocks) * esz)

    def feed_begin(i):
        runner.feed_resident_begin(blocks.data_ptr() + (i % nblocks) * esz)

    def skip(i):
        runner.count += 1

    def barrier():
        dist.barrier()
        torch.cuda.synchronize(dev)
    from pycusdr_amd.dist import StepWatchdog
    dog = StepWatchdog(args.watchdog, rank=rank, describe=shard.describe) if args.watchdog > 0 else None
    shard.run(runner, range((1 + args.warmup) * G), decoder=decoder, feed=feed, skip=skip, watchdog=dog, feed_begin=feed_begin)
    barrier()
    runner.demod.bank.profile_enable(True)
    t0 = time.perf_counter()
    res, packets = shard.run(runner, range(args.steps * G), decoder=decoder, feed=feed, skip=skip, watchdog=dog, feed_begin=feed_begin)
    barrier()
    if dog is not None:
        dog.stop()
    elapsed = time.perf_counter() - t0
    counts, kms = runner.demod.bank.profile_read()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    pinfo = runner.demod.bank.get_search_path()
    Mu = runner.demod.bank.get_info()[2]
    props = torch.cuda.get_device_properties(dev)
    rank_devices = [None] * G
    dist.all_gather_object(rank_devices, {'rank': rank, 'device': local_rank, 'uuid': str(getattr(props, 'uuid', '')), 'name': props.name})
    if rank == 0:
        nb = args.steps * G
        host_ms = float(np.mean([d['time_ms'] for d in res])) if res else None
        out = {'metric': 'IQ Msamples/sec through Doppler matched-filter bank (256 bins, 2^20 chunk)',
               'value': round(nb * (N - ov) / elapsed / 1e6, 3), 'unit': 'Msamples/s', 'n_gpus': G, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'block round-robin (BlockShard): each of {G} GPUs runs the full D={args.bins} bank of {args.protocol} '
                                      f'(N=2^{log2N}) on every {G}th block -- search, pick, matched filters at the found shift, symbol '
                                      f'decisions, bit lookup and alignment (the previous block\'s tail comes from its owner) --, rank 0 '
                                      f'runs the decoder in block order; blocks resident in HBM, two point-to-point messages per block, no collective',
                          'shard': 'blocks', 'world_size': G, 'backend': args.backend, 'blocks_timed': nb, 'path': pinfo,
                          'rccl_world': dist.get_world_size() if args.backend == 'nccl' else None, 'rank_devices': rank_devices,
                          'distinct_devices': len({(d['device'], d['uuid']) for d in rank_devices}),
                          'packets_found': len(packets), 'mean_block_ms_device_plus_host_on_root': host_ms,
                          'root_ms_per_block': round(shard.stats['root_s'] / max(shard.stats['root_blocks'], 1) * 1e3, 4),
                          'root_wait_ms_per_block': round(shard.stats['root_wait_s'] / max(shard.stats['root_blocks'], 1) * 1e3, 4),
                          'owner_host_ms_per_own_block': round(shard.stats['host_s'] / max(shard.stats['own_blocks'], 1) * 1e3, 4),
                          'root_note': 'root_ms_per_block: result dict + decoder, the only work the root does for EVERY block (root_wait_ms_per_block: '
                                       'time it sat in the receive of a block that had not arrived yet); '
                                       'owner_host_ms_per_own_block: tail exchange + bit lookup + alignment + hand-over, done by each owner for its own blocks',
                          'units': 'samples of the one physical stream (every block is processed once)'},
               'roofline': segment_roofline_core(pinfo, args.bins, Mu, counts[0], kms[0]) if pinfo['path'] == 'segment' else None,
               'cpu_baseline': None}
        print(json.dumps(out), flush=True)
    runner.close()
    dist.barrier()
    dist.destroy_process_group()


def run_blocks_leg(args, dist, rank, G, local_rank, dev, steps=6, warmup=2):
    """After the bins-mode loop of an N > 1 job: a short leg of the OTHER sharding axis of SURVEY 8(e) in the same job -- every GPU
    runs the full 256-bin bank on every G-th time block (pycusdr_amd.dist.BlockShard, no collective on the data path) -- and,
    for scale, the same loop on rank 0 alone.  BASELINE's wording (256 bins at 1/2/4/8 GPUs) is this axis.  Returns flat keys
    on rank 0, None elsewhere."""
    import torch
    from pycusdr_amd import config as cfg, signals as sg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.dist import BlockShard
    from pycusdr_amd.protocol import loadProtocol
    log2N, ov = args.log2n, 1 << 10
    N = 1 << log2N
    conf = cfg.bench_config(args.protocol, blockSize=log2N, overlap=10, doppCarrierSteps=args.bins, device=local_rank)
    rr, _ = widen_range_rate(conf, 'UHF-H', N, args.bins)
    conf['Radios']['rangeRateMax'] = rr
    proto = loadProtocol(args.protocol)(conf=conf)
    runner = DemodulatorRunner(conf, proto, 'UHF-H')
    group = dist.new_group(backend='gloo') if args.backend == 'nccl' else None     # the hand-back moves host arrays
    solo = dist.new_group(ranks=[0], backend='gloo')                                # (collective: every rank calls it)
    nblocks = 8
    stream = sg.s1_stream(nblocks, N, ov, 'GMSK', 16, 153600, snr_db=10.0, seed=1)
    host_blocks = np.stack([stream[b * (N - ov): b * (N - ov) + N] for b in range(nblocks)])
    blocks = torch.from_numpy(host_blocks.view(np.float32).reshape(nblocks, 2 * N)).to(dev)
    esz = blocks.element_size() * 2 * N
    torch.cuda.synchronize(dev)
    decoder = Decoder(conf, proto) if rank == 0 else None

    def feed(i):
        return runner.feed_resident(blocks.data_ptr() + (i % nblocks) * esz)

    def feed_begin(i):
        runner.feed_resident_begin(blocks.data_ptr() + (i % nblocks) * esz)

    def skip(i):
        runner.count += 1

    # a hang in this leg must not cost the job its line: the watchdog turns it into exit status 3, and the fallback attempt of the
    # launcher ladder runs without the leg (SAFE_MODE carries --no-blocks-leg)
    from pycusdr_amd.dist import StepWatchdog
    dog = StepWatchdog(args.watchdog, rank=rank, describe=lambda: 'time-chunk-sharded leg (run_blocks_leg)') if args.watchdog > 0 else None

    def measure(shard, members, n):
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if members:
            shard.run(runner, range(n), decoder=decoder, feed=feed, skip=skip, feed_begin=feed_begin, watchdog=dog)
        if dog is not None:
            dog.beat()
        torch.cuda.synchronize(dev)
        dist.barrier()
        if dog is not None:
            dog.beat()
        return time.perf_counter() - t0
    shard = BlockShard(group=group)
    measure(shard, True, warmup * G)
    t_all = measure(shard, True, steps * G)
    one = BlockShard(rank=0, world=1, group=solo) if rank == 0 else None
    measure(one, rank == 0, warmup)
    t_one = measure(one, rank == 0, steps)
    runner.close()
    if dog is not None:
        dog.stop()
    if rank != 0:
        return None
    all_ms, one_ms = steps * G * (N - ov) / t_all / 1e6, steps * (N - ov) / t_one / 1e6
    return {'blocks_stream_msamples': round(all_ms, 2), 'blocks_1gpu_msamples': round(one_ms, 2),
            'blocks_efficiency_vs_1gpu': round(all_ms / (G * one_ms), 4), 'blocks_leg_steps': steps}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


SAFE_MODE = ['--single-comm', '--no-prefetch', '--no-blocks-leg']


def _run_child(cmd, env):
    """Start `cmd` as a child process, pass SIGTERM / SIGINT on to it, return its exit code."""
    import signal
    import subprocess
    child = subprocess.Popen(cmd, env=env)

    def forward(sig, _frame):
        try:
            child.send_signal(sig)
        except OSError:
            pass
    old = {sg_: signal.signal(sg_, forward) for sg_ in (signal.SIGTERM, signal.SIGINT)}
    try:
        return child.wait()
    finally:
        for sg_, h in old.items():
            signal.signal(sg_, h)


def spawn_ranks(n, backend, argv):
    """`python bench.py --gpus N` without a launcher: run the documented launch line as a child process
    (python -m torch.distributed.run, one fresh rank per GPU) and return its exit code.  With the RCCL backend the
    node must have N devices -- a smaller box is an error, never a smaller figure.

    One-shot safety: the first attempt runs the default mode (block broadcast on a communicator and stream of its own, one
    block ahead) under a 60 s watchdog; if it leaves with a non-zero status -- a collective some rank never joined, named by
    the watchdog -- a FRESH set of ranks is started in the conservative mode (--single-comm --no-prefetch: every collective of
    the job in one program order on one stream).  This process never touches the GPU; no rank is ever re-executed.  The line
    says which mode produced it (config.dist_mode, config.fallback_from)."""
    have = kfd_gpu_count()
    if have is None:
        import torch
        have = torch.cuda.device_count()
    if backend == 'nccl' and have < n:
        sys.stderr.write(f'bench.py: --gpus {n} needs {n} GPUs, this node has {have} (use --backend gloo to rehearse '
                         f'{n} ranks on fewer devices)\n')
        return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['BENCH_WORKER'] = '1'            # the ranks below are the workers themselves: no second layer of children

    def launch(extra):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
               '--master-port', str(free_port()), os.path.abspath(__file__)] + argv + extra
        sys.stderr.write('bench.py: no launcher (WORLD_SIZE unset), starting ' + ' '.join(cmd) + '\n')
        return _run_child(cmd, env)
    conservative = all(a in argv for a in SAFE_MODE[:2])
    rc = launch([] if conservative or '--watchdog' in argv else ['--watchdog', '60'])
    if rc == 0 or conservative or '--shard' in argv and argv[argv.index('--shard') + 1] == 'blocks':
        return rc
    sys.stderr.write(f'bench.py: the ranks left with status {rc} in the default mode; starting fresh ranks with {" ".join(SAFE_MODE)}\n')
    env.pop('BENCH_FAIL_FIRST_ATTEMPT', None)           # (test knob: only the first attempt is made to fail)
    return launch(SAFE_MODE + ['--fallback-from', f'default mode (concurrent broadcast, prefetch) exit {rc}'])


def rank_supervisor(argv):
    """Started BY a launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`) with N > 1: this rank process
    does not touch the GPU either -- it runs the real worker as a child (same environment, BENCH_WORKER=1) under a 60 s
    watchdog and, if the worker leaves with a non-zero status in the default mode, starts a fresh worker in the conservative
    mode on a rendezvous of its own (MASTER_PORT + 1, rank 0 hosts the store).  Every rank's worker fails together -- a hung
    collective stops all of them, each watchdog fires -- so every supervisor takes the same decision."""
    env = dict(os.environ)
    env['BENCH_WORKER'] = '1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    me = [sys.executable, os.path.abspath(__file__)]
    conservative = all(a in argv for a in SAFE_MODE[:2])
    rc = _run_child(me + argv + ([] if conservative or '--watchdog' in argv else ['--watchdog', '60']), env)
    if rc == 0 or conservative or '--shard' in argv and argv[argv.index('--shard') + 1] == 'blocks':
        return rc
    sys.stderr.write(f'bench.py: rank {env.get("RANK")}: the worker left with status {rc} in the default mode; starting a fresh worker '
                     f'with {" ".join(SAFE_MODE)}\n')
    env['MASTER_PORT'] = str(int(env.get('MASTER_PORT', '29500')) + 1)
    env.pop('TORCHELASTIC_USE_AGENT_STORE', None)        # the launcher's store holds the first attempt's keys: rendezvous afresh
    env.pop('BENCH_FAIL_FIRST_ATTEMPT', None)
    return _run_child(me + argv + SAFE_MODE + ['--fallback-from', f'default mode (concurrent broadcast, prefetch) exit {rc}'], env)


def main():
    from pycusdr_amd.hostcpu import quiet_blas
    quiet_blas()         # numpy's BLAS workers spinning under a CPU quota stall the whole process (pycusdr_amd/hostcpu.py)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=25,
                    help='untimed steps before the clock starts; whatever the value, untimed steps go on until 50 ms of work have '
                         'passed: after an idle spell the device needs ~30 ms to settle its clock (tools/ramp_probe.py)')
    ap.add_argument('--repeats', type=int, default=5,
                    help='the --steps loop is timed this many times back to back; value / ms_per_step are the median repeat, '
                         'config.ms_per_step_min / _max the extremes')
    ap.add_argument('--log2n', type=int, default=20)
    ap.add_argument('--bins', type=int, default=256, help='Doppler bins per GPU')
    ap.add_argument('--protocol', default='bench_GMSK')
    ap.add_argument('--signal', choices=['S1', 'S2'], default='S1',
                    help='stimulus of the timed loop: S1 = GMSK bench packet at +fs/4 with AWGN 10 dB (default), S2 = white noise '
                         '(the default run times S2 as well, config.s2_*; this switch is for profile runs of the whole line on S2)')
    ap.add_argument('--path', choices=['auto', 'segment', 'twopass'], default='auto')
    ap.add_argument('--seg', default='', help='segment path tuning: log2L,wg_per_cu,filters_per_pass (0 = default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the untimed secondary figures (demodulation leg, other path, S2, sync correlator): profiling runs')
    ap.add_argument('--tuning', default='', help='two-pass path: chunk,mpb,rows,jsplit (0 = default)')
    ap.add_argument('--shard', choices=['bins', 'blocks'], default='bins',
                    help='N>1: bins = C4, Doppler bins sharded 256/GPU + RCCL exchange per block (default); '
                         'blocks = every GPU runs the full 256-bin bank on different time blocks, no collective')
    ap.add_argument('--no-prefetch', action='store_true', help='N>1: broadcast every block right before its search instead of one block ahead')
    ap.add_argument('--single-comm', action='store_true',
                    help='N>1: the block broadcast shares the communicator and the stream of the score exchange (one program order of '
                         'collectives per rank) instead of running beside the search on a communicator of its own')
    ap.add_argument('--watchdog', type=float, default=120.0,
                    help='N>1 / --force-dist: seconds without a completed step after which a rank prints what it is stuck in and '
                         'exits with status 3 (0 disables)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="process-group backend; 'gloo' lets several ranks share ONE GPU (rehearsal of the N>1 path on a 1-GPU box)")
    ap.add_argument('--force-dist', action='store_true', help='run the sharded/RCCL path even with one rank (rehearsal)')
    ap.add_argument('--fallback-from', default=None, help='(set by the launcher ladder) the mode whose failure led to this run')
    ap.add_argument('--no-blocks-leg', action='store_true', help='N>1: skip the short time-chunk-sharded leg after the bins-mode loop')
    ap.add_argument('--no-chain', action='store_true', help='N=1: skip the receive-chain figures at the reference block sizes')
    ap.add_argument('--no-c5', action='store_true', help='N=1: skip the two-concurrent-instances leg (BASELINE C5: CC11xx + BPSK at 512 bins)')
    ap.add_argument('--no-other-banks', action='store_true',
                    help='skip the untimed per-bank figures (CC11xx sps 128, BPSK M=32 at D=256; C3 D=1024) in config.other_banks / config.c3')
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is not None and int(env_world) != args.gpus and not (args.force_dist and args.gpus == 1 and int(env_world) == 1):
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; refusing to print a '
                         f'{env_world}-rank figure as the {args.gpus}-GPU result\n')
        sys.exit(2)
    if args.gpus > 1 and env_world is None:
        # started without a launcher: start one rank per GPU as FRESH child processes (nothing in this process has
        # touched the GPU: the device count comes from the kernel driver's topology files) and leave with their exit code
        sys.exit(spawn_ranks(args.gpus, args.backend, sys.argv[1:]))
    if args.gpus > 1 and os.environ.get('BENCH_WORKER') != '1':
        # started by a launcher: supervise the real worker from here (one-shot safety, see rank_supervisor)
        sys.exit(rank_supervisor(sys.argv[1:]))

    import torch
    import __graft_entry__
    __graft_entry__.build()
    from pycusdr_amd import config as cfg, signals as sg
    from pycusdr_amd.mfbank import MFBank
    from pycusdr_amd.protocol import loadProtocol
    from pycusdr_amd.dist import DopplerShard, bin_slice

    G = args.gpus
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if G > 1 or world > 1 or args.force_dist:
        import torch.distributed as dist
        if 'MASTER_PORT' not in os.environ:          # --force-dist rehearsal with one rank only
            os.environ['MASTER_PORT'] = str(free_port())
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'gloo':                      # rehearsal: ranks may share a device
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        # peers that never arrive must cost a worker 90 s, not the 10-minute default (the driver's whole budget): it then exits
        # non-zero in time for the supervisor's fallback attempt to print a line
        from datetime import timedelta
        pg_timeout = timedelta(seconds=float(os.environ.get('BENCH_PG_TIMEOUT_S', '90')))
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group('gloo', timeout=pg_timeout)
        if dist.get_world_size() != G:       # never report a smaller world under the requested --gpus
            sys.stderr.write(f'bench.py: --gpus {G} but the process group has {dist.get_world_size()} ranks\n')
            sys.exit(2)
        if rank == 0:
            sys.stderr.write(f'bench.py: world_size {dist.get_world_size()} == --gpus {G}, backend {args.backend}\n')
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank)
    if args.shard == 'blocks' and dist is not None:
        return run_block_shard(args, dist, rank, G, local_rank, dev)

    log2N, ov = args.log2n, 1 << 10
    N = 1 << log2N
    by_blocks = args.shard == 'blocks'
    D_total = args.bins if by_blocks else args.bins * G
    M_size = 5 if args.protocol == 'bench_BPSK' else 3
    if args.protocol == 'CC11xx':        # config/CC11xx.json geometry: FSK-2 at 128 samples per symbol (384-tap filters)
        conf, sps_bank = cfg.cc11xx_config(blockSize=log2N, overlap=10, doppCarrierSteps=D_total, device=local_rank), 128
    else:
        conf, sps_bank = cfg.bench_config(args.protocol, blockSize=log2N, overlap=10, doppCarrierSteps=D_total, device=local_rank), 16
    rr, shifts = widen_range_rate(conf, 'UHF-H', N, D_total)
    conf['Radios']['rangeRateMax'] = rr
    proto = loadProtocol(args.protocol)(conf=conf)
    M, masks = proto.get_filter(N, sps_bank, M_size)
    lo, hi = (0, D_total) if by_blocks else bin_slice(D_total, rank, G)

    bank = MFBank(log2N, hi - lo, M, window_width=7, sum_all_masks=True, device=local_rank)
    bank.set_filters(masks)
    bank.set_shifts(shifts[lo:hi])
    seg = [int(v) for v in args.seg.split(',')] if args.seg else []
    if args.path != 'auto' or seg:
        bank.set_search_path(args.path if args.path != 'auto' or not seg else 'segment', *seg)
    if args.tuning:
        bank.set_tuning(*[int(v) for v in args.tuning.split(',')])
    pinfo = bank.get_search_path()
    sinfo = bank.get_search_info()
    shard = None
    if (G > 1 or args.force_dist) and not by_blocks:
        shard = DopplerShard(rank=rank, world=G, device=dev, concurrent_broadcast=not args.single_comm)
        shard.attach(bank, D_total, M, sum_all=True)

    # synthetic input S1: the reference's GMSK bench packet at +fs/4, tiled, AWGN 10 dB, resident in HBM
    nblocks = 16
    stream = sg.s1_stream(nblocks, N, ov, 'GMSK', 16, 153600, snr_db=10.0, seed=1)
    host_blocks = np.stack([stream[b * (N - ov): b * (N - ov) + N] for b in range(nblocks)])
    blocks = torch.from_numpy(host_blocks.view(np.float32).reshape(nblocks, 2 * N)).to(dev)
    esz = blocks.element_size() * 2 * N
    # S2 (unit-variance white noise, what a receiver sees between passes; the reference's bench adds exactly such
    # noise, examples/benchmark/create_signals.py:115-141): sixteen blocks resident in HBM as well, generated BEFORE the warm-up --
    # half a second of host work with an idle device in front of a timed loop would put that loop into the clock ramp
    s2_blocks = None
    if G == 1 and not args.force_dist and not args.no_extras or args.signal == 'S2':
        s2_blocks = torch.from_numpy(sg.s2_noise(nblocks, N).view(np.float32).reshape(nblocks, 2 * N)).to(dev)
    if args.signal == 'S2':            # profiling runs: the whole line on S2
        blocks, host_blocks = s2_blocks, s2_blocks.cpu().numpy().view(np.complex64).reshape(nblocks, N)
    torch.cuda.synchronize(dev)        # the blocks are resident before any other stream reads them

    def block_index(i):
        return (i * (G if by_blocks else 1) + (rank if by_blocks else 0)) % nblocks

    def step(i, src=None):
        src = blocks if src is None else src
        if shard is None:
            bank.upload_device(src.data_ptr() + block_index(i) * esz)
            return bank.find_carrier()
        # sharded: rank 0 owns the stream; its block goes to every rank over RCCL (the next block's broadcast is
        # started beside this block's search), then search + exchange + pick
        return shard.step(bank, lo, src[block_index(i)] if rank == 0 else None,
                          next_block=src[block_index(i + 1)] if rank == 0 else None, prefetch_next=not args.no_prefetch)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if shard is not None and args.watchdog > 0:
        # a collective some rank never joins waits for ever: say which one and leave with a non-zero status instead
        from pycusdr_amd.dist import StepWatchdog
        dog = StepWatchdog(args.watchdog, rank=rank, describe=shard.describe)
        plain_step = step

        def step(i, src=None):              # noqa: F811
            out_ = plain_step(i, src)
            dog.beat(i)
            return out_

    if os.environ.get('BENCH_FAIL_FIRST_ATTEMPT') and shard is not None and not args.single_comm and rank == G - 1:
        # test knob: the last rank of the first (default-mode) attempt never joins the first collective -- the others sit in
        # it until their watchdog names it and exits 3; this rank leaves the same way a little later
        time.sleep(args.watchdog + 2.0)
        os._exit(3)
    step(0)            # initialisation: first launches load the code objects and touch the workspaces
    torch.cuda.synchronize(dev)
    t_w = time.perf_counter()
    for i in range(1, 1 + args.warmup):     # one running block counter: a prefetched block is always the next one used
        res = step(i)
    # ... and keep going, untimed, until at least SETTLE_S of work have passed since the initialisation, whatever --warmup says:
    # after an idle spell the device runs its first ~30 ms 8-20 % slow (profiles/r03_ramp.md), and a receiver on a
    # continuous stream is never there.  Every rank runs the same number of steps (the sharded step is collective).
    first = 1 + args.warmup
    while True:
        busy = torch.tensor([time.perf_counter() - t_w], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(busy, op=dist.ReduceOp.MIN)
        if float(busy.item()) >= SETTLE_S and first > 1:
            break
        for _ in range(8):
            res = step(first)
            first += 1
    # the clock: EXACTLY --steps steps between barrier + synchronize on both sides, maximum over ranks -- taken args.repeats times
    # back to back; the line reports the median repeat and the extremes beside it
    bank.profile_enable(True)
    bank.timer_start()
    times = []
    for rep in range(args.repeats):
        barrier()
        t0 = time.perf_counter()
        for i in range(first, first + args.steps):
            res = step(i)
        barrier()
        dt_rep = time.perf_counter() - t0
        first += args.steps
        if dist is not None:
            t = torch.tensor([dt_rep], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_rep = float(t.item())
        times.append(dt_rep)
    ev_ms = bank.timer_stop()
    counts, kms = bank.profile_read()
    bank.profile_enable(False)
    if shard is not None and args.watchdog > 0:
        dog.stop()
    elapsed = float(np.median(times))
    timed_steps = args.steps * args.repeats
    # scores of the LAST timed block, read before anything else touches the handle (parity spot check)
    last_block = block_index(first - 1)
    gscores = bank.get_scores()[:, 0].astype(np.float64)

    # ---- S2: the SAME loop on white noise (same settle rule, --steps, --repeats, sixteen blocks resident) --------------------
    s2 = None
    if s2_blocks is not None and shard is None and args.signal == 'S1':
        t_w, i2 = time.perf_counter(), 0
        while time.perf_counter() - t_w < SETTLE_S or i2 < 8:
            for _ in range(8):
                step(i2, s2_blocks)
                i2 += 1
        bank.profile_enable(True)
        times2 = []
        for rep in range(args.repeats):
            barrier()
            t0 = time.perf_counter()
            for i in range(i2, i2 + args.steps):
                step(i, s2_blocks)
            barrier()
            times2.append(time.perf_counter() - t0)
            i2 += args.steps
        c2, k2 = bank.profile_read()
        bank.profile_enable(False)
        s2 = {'times': times2, 'counts': c2, 'kms': k2, 'untimed_steps_before': i2 - args.steps * args.repeats}

    # live sanity: the pick must land on the +fs/4 carrier
    frac_idx = float(res[0])
    pick_shift = float(np.interp(frac_idx, np.arange(D_total), np.where(shifts > N // 2, shifts - N, shifts)))
    spacing = float(np.median(np.diff(np.sort(shifts))))
    # (only the GMSK bank matches the S1 stimulus; the other banks are timed on it for throughput alone)
    carrier_ok = abs(pick_shift - N / 4) <= 1.5 * spacing if args.protocol == 'bench_GMSK' and args.signal == 'S1' else None

    # ---- untimed secondary figures -----------------------------------------------------
    extras = {}
    other = None
    if shard is None and not args.no_extras:
        # find_carrier + demodulate (A3..A11 device part), host arithmetic as Demodulator.findCodeRateAndPhaseGPU
        k_off = int(N / (1.1 * 16))
        k_len = int(N / (0.9 * 16)) - k_off
        reps = 5
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(reps):
            step(i)
            k, arg, _ = bank.demodulate(N // 4, k_off, k_len)
            spS = N / float(k) if float(k) else 10.0
            cOff = -float(arg) / np.pi * spS / 2
            if cOff < 0:
                cOff += spS - 1
            bank.find_centres(np.float32(spS), np.float32(cOff), 0, int(N / spS))
        torch.cuda.synchronize(dev)
        extras['find_carrier_plus_demodulate_ms'] = round((time.perf_counter() - t1) / reps * 1e3, 4)
        # the same stages (plus the shift interpolation, the SNR windows and the rate arithmetic) as ONE library call with one
        # synchronisation: mfb_receive_block on the block resident in HBM
        bank.receive_block(k_off, k_len, 8, source='device', device_ptr=blocks.data_ptr())
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(reps):
            bank.receive_block(k_off, k_len, 8, source='device', device_ptr=blocks.data_ptr() + block_index(i) * esz)
        extras['receive_block_one_call_ms'] = round((time.perf_counter() - t1) / reps * 1e3, 4)
        # the same call as begin / end with two blocks in flight (the next block's launches are enqueued before this block's
        # results are waited for), on one stream and -- mfb_set_batch_overlap -- as two parts on two streams: the next block's search
        # beside this block's matched filters, envelope transform, rate, centres and read-back (Next 8)
        def begin_end_loop(n):
            bank.begin_block(0, k_off, k_len, 8, source='device', device_ptr=blocks.data_ptr())
            for k in range(1, n):
                bank.begin_block(k & 1, k_off, k_len, 8, source='device', device_ptr=blocks.data_ptr() + block_index(k) * esz)
                bank.end_block((k - 1) & 1)
            bank.end_block((n - 1) & 1)
        for key, on in (('receive_block_begin_end_ms', False), ('receive_block_pipelined_ms', True)):
            try:
                bank.set_batch_overlap(on)
                begin_end_loop(8)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                begin_end_loop(24)
                extras[key] = round((time.perf_counter() - t1) / 24 * 1e3, 4)
            except Exception as e:       # noqa: BLE001 -- a secondary figure must not cost the line
                extras[key + '_error'] = str(e)[:100]
            finally:
                bank.set_batch_overlap(False)
        # the other search path, same blocks, a few steps (its own roofline accounting)
        if G == 1:
            try:
                bank.set_search_path('twopass' if pinfo['path'] == 'segment' else 'segment')
                o_info = bank.get_search_path()
                step(0)
                torch.cuda.synchronize(dev)
                bank.profile_enable(True)
                t1 = time.perf_counter()
                osteps = 4
                for i in range(osteps):
                    step(i)
                torch.cuda.synchronize(dev)
                o_el = time.perf_counter() - t1
                o_counts, o_kms = bank.profile_read()
                bank.profile_enable(False)
                other = (o_info, o_el / osteps, o_counts, o_kms, bank.get_tuning(), osteps, bank.get_search_info())
            except ValueError:
                other = None
            bank.set_search_path(args.path if args.path != 'auto' or not seg else 'segment', *seg)
        # opt-in span basis of the SUM_ALL search (rank(bank) filters instead of M) -- never the headline
        if G == 1 and pinfo['path'] == 'segment':
            try:
                bank.set_search_basis('span')
                sb = bank.get_search_basis()
                step(0)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for i in range(8):
                    step(i)
                torch.cuda.synchronize(dev)
                dt_sb = (time.perf_counter() - t1) / 8
                sc_sb = bank.get_scores()[:, 0].astype(np.float64)
                bank.set_search_basis('filters')
                step(7)
                sc_f = bank.get_scores()[:, 0].astype(np.float64)
                extras['span_basis_search'] = {'filters_transformed': sb[1], 'of': M, 'ms_per_step': round(dt_sb * 1e3, 4),
                                               'msamples': round((N - ov) / dt_sb / 1e6, 2),
                                               'max_rel_diff_vs_default_search': float(np.abs(sc_sb - sc_f).max() / sc_f.max()),
                                               'note': 'opt-in (mfb_set_search_basis): exact identity for the SUM_ALL_MASKS score; not the headline'}
            except (ValueError, RuntimeError):
                bank.set_search_basis('filters')
        # opt-in spectral-energy search (Parseval: no inverse transform at all) -- never the headline
        if G == 1:
            try:
                bank.set_search_mode('energy')
                step(0)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for i in range(32):
                    step(i)
                torch.cuda.synchronize(dev)
                dt_en = (time.perf_counter() - t1) / 32
                sc_en = bank.get_scores()[:, 0].astype(np.float64)
                bank.set_search_mode('transforms')
                step(31)
                sc_f = bank.get_scores()[:, 0].astype(np.float64)
                extras['energy_search'] = {'ms_per_step': round(dt_en * 1e3, 4), 'msamples': round((N - ov) / dt_en / 1e6, 2),
                                           'max_rel_diff_vs_default_search': float(np.abs(sc_en - sc_f).max() / sc_f.max()),
                                           'note': 'opt-in (mfb_set_search_mode): Parseval identity on |X|^2 and the filters\' energy, D*N '
                                                   'multiply-adds, the matched-filter bank does not run; not the headline'}
            finally:
                bank.set_search_mode('transforms')
        # sync/preamble correlator (A14): B = 1024 streams of 67 584 bits x 64 taps, thresholded on the device.  The streams
        # travel PACKED (8 bits per byte, np.packbits layout) from the library's page-locked buffer; only the hits come back.
        from pycusdr_amd.mfbank import sync_find_packed, sync_pinned_buffer
        rsb = np.random.RandomState(2)
        B, Lb = 1024, 65536 + 2048
        bits = rsb.randint(0, 2, (B, Lb)).astype(np.uint8)
        tmpl = (2 * rsb.randint(0, 2, 64) - 1).astype(np.int8)
        header = ((tmpl[::-1] + 1) // 2).astype(np.uint8)
        for pos in range(100, Lb - 64, 4000):
            bits[:, pos:pos + 64] = header
        thr = int((tmpl == 1).sum()) - 5                      # numOnes - tolerance, as decoder.py:101
        row_bytes = (Lb + 7) // 8
        stage = sync_pinned_buffer(B * row_bytes, device=local_rank)[:B * row_bytes].reshape(B, row_bytes)
        stage[:] = np.packbits(bits, axis=1)
        sync_find_packed(stage, Lb, tmpl, thr, device=local_rank)       # warm-up (allocations, code objects)
        reps, dts, dev_ms = 5, [], []
        for _ in range(reps):
            t1 = time.perf_counter()
            (hcnt, hidx, hsc), kms_sync = sync_find_packed(stage, Lb, tmpl, thr, device=local_rank, timing=True, flat=True)
            dts.append(time.perf_counter() - t1)
            dev_ms.append(kms_sync)
        dt, kdev = float(np.median(dts)), float(np.median(dev_ms))
        nh = int(hcnt.sum())
        hits = [(hidx[:hcnt[0]], hsc[:hcnt[0]])]
        bytes_in, bytes_out = B * row_bytes, nh * 8 + B * 4
        ref0 = np.convolve(bits[0].astype(np.int64), tmpl.astype(np.int64))
        extras['sync_correlator'] = {
            'streams_per_s': round(B / dt, 1), 'B': B, 'bits_per_stream': Lb, 'taps': 64, 'hits_per_stream': int(len(hits[0][0])),
            'layout': 'packed bits (np.packbits), page-locked staging; XOR/AND + popcount on 64-bit windows; hits only back',
            'call_ms': round(dt * 1e3, 4), 'device_ms': round(kdev, 4), 'bytes_in': bytes_in, 'bytes_out': bytes_out,
            'pcie_frac_of_63GBps': round((bytes_in + bytes_out) / dt / 63e9, 4),
            'device_hbm_frac_of_8TBps': round((bytes_in + bytes_out) / (kdev * 1e-3) / HBM_PEAK, 4) if kdev > 0 else None,
            'device_note': 'one sweep over the packed streams (the second pass revisits only segments that hold a hit): '
                           '(bytes_in + bytes_out) over device_ms; the sweep is instruction-bound, the call is bound by the host link',
            'exact_vs_np_convolve_stream0': bool(np.array_equal(hits[0][0], np.where(ref0 >= thr)[0])
                                                 and np.array_equal(hits[0][1], ref0[ref0 >= thr])),
            'includes': 'H2D of the packed streams from page-locked memory, kernels, D2H of the hits'}

    if shard is None and G == 1 and not args.no_extras and not args.no_other_banks:
        # the other BASELINE-named banks at the same geometry (C5's two filter sets at D = 256) and C3 (D = 1024)
        extras['other_banks'] = [bank_figure(dev, local_rank, p, args.bins, log2N, blocks, esz, nblocks, s2_blocks=s2_blocks, span=True)
                                 for p in ('CC11xx', 'bench_BPSK') if p != args.protocol]
        extras['c3'] = bank_figure(dev, local_rank, args.protocol, 1024, log2N, blocks, esz, nblocks, steps=4, warmup=1, twopass_steps=3)

    c5 = None
    if shard is None and G == 1 and not args.no_extras and not args.no_c5 and rank == 0 and log2N == 20 and args.bins == 256:
        c5 = c5_inprocess_figures(dev, local_rank, blocks, esz, nblocks)
        c5.update(c5_concurrent_figures())
    blocks_leg = None
    if dist is not None and G > 1 and shard is not None and not args.no_blocks_leg:
        blocks_leg = run_blocks_leg(args, dist, rank, G, local_rank, dev)
    chain = None
    if shard is None and G == 1 and not args.no_extras and not args.no_chain and rank == 0 and log2N == 20 and args.bins == 256:
        chain = chain_figures(local_rank)           # (the default run only: C2's line carries the chain at the reference's own block sizes)

    # who took part: one entry per rank (process rank, device index, device uuid / name), gathered over the communicator
    props = torch.cuda.get_device_properties(dev)
    me = {'rank': rank, 'device': local_rank, 'uuid': str(getattr(props, 'uuid', '')), 'name': props.name}
    rank_devices = [me]
    if dist is not None:
        rank_devices = [None] * dist.get_world_size()
        dist.all_gather_object(rank_devices, me)

    out = None
    rc = 0
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (N - ov) * G / (elapsed / args.steps) / 1e6
        Dl = hi - lo
        tun = bank.get_tuning()
        Mu = bank.get_info()[2]          # filter rows the search really transforms (exact duplicates/negatives once)
        t_block_dev = ev_ms / timed_steps * 1e-3

        def twopass_roofline(Dl_, counts_, kms_, tun_, nsteps):
            dom = 0 if kms_[0] >= kms_[1] else 1
            names = ['k_pass1<256,BANK> (shift-multiply + column FFT + twiddle -> Z)',
                     'k_pass2<4096,REDUCE> (row FFT + |.|^2 reduction)']
            launches = max(counts_[dom], 1)
            bins_per_launch = Dl_ * nsteps / launches
            if dom == 0:
                k_bytes = 8.0 * bins_per_launch * Mu * N + 8.0 * N * (1 + Mu)   # Z write + spectrum + filter bank read once
            else:
                k_bytes = 8.0 * bins_per_launch * Mu * N + 4.0 * bins_per_launch * Mu  # Z read + partial sums
            k_avg_s = kms_[dom] / launches * 1e-3
            traffic, tsrc = None, None
            tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(tfile):
                try:
                    rec = json.load(open(tfile)).get(f'D{Dl_}_M{M}_N{log2N}_chunk{tun_[0]}', {})
                    traffic, tsrc = rec.get('pass1' if dom == 0 else 'pass2'), rec.get('source')
                except Exception:
                    traffic, tsrc = None, None
            return {'bound': 'hbm', 'kernel': names[dom], 'achieved': round(k_bytes / k_avg_s / 1e9, 2), 'peak': HBM_PEAK / 1e9,
                    'unit': 'GB/s', 'frac': round(k_bytes / k_avg_s / HBM_PEAK, 4), 'traffic': traffic, 'traffic_source': tsrc,
                    'traffic_over_alg': round(traffic / k_bytes, 3) if traffic else None,
                    'launches': launches, 'avg_launch_ms': round(k_avg_s * 1e3, 4), 'alg_bytes_per_launch': k_bytes,
                    'other_kernel_avg_ms': round(kms_[1 - dom] / max(counts_[1 - dom], 1), 4)}

        def segment_roofline(info, Dl_, counts_, kms_, sinfo_=None):
            L, V = 1 << info['log2L'], info['valid_per_segment']
            core = segment_roofline_core(info, Dl_, Mu, counts_[0], kms_[0], sinfo_ if sinfo_ is not None else sinfo)
            traffic, tsrc = None, None
            tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(tfile):
                try:
                    rec = json.load(open(tfile)).get(f'segment_D{Dl_}_M{M}_N{log2N}_L{info["log2L"]}', {})
                    traffic, tsrc = rec.get('bytes'), rec.get('source')
                except Exception:
                    pass
            fs_ = core['filter_side_shift']
            core['kernel'] = (f"{'k_segf' if fs_ else 'k_seg'}<{L},REDUCE,{V // (L // (32 if L == 2048 else 16))}> (" +
                              (f"one forward FFT per segment and {core['bins_per_forward']} bins; per (bin, filter): product with that bin's "
                               f"segment spectra, inverse FFT, |.|^2 sums" if fs_ else
                               'mix + forward FFT per (bin, segment); per filter: product, inverse FFT, |.|^2 sums') +
                              f"; segments of {L} points, {V} valid outputs, {info['segments']} segments per bin)")
            core.update({
                'traffic': traffic, 'traffic_source': tsrc,
                'flops_formula': ('D*Q*Mu*(6L + 5L log2 L + 4V) + Q*ceil(D/bins_per_forward)*5L log2 L' if fs_ else
                                  'D*Q*((6L + 5L log2 L) + Mu*(6L + 5L log2 L + 4V))'),
                'frac_note': 'frac counts the transforms THIS kernel runs at the nominal 5 L log2 L each (the forward transform of a segment is '
                             'shared by bins_per_forward bins and there is no mixing multiply when the shift sits on the filters; what a transform '
                             'prunes -- dead output slots since round 2, all but the invalid outputs of the second pass in round 6, whose valid '
                             'energy is the total by Parseval minus the invalid part, the total once per bin from a table of sum_f |G_f|^2 in SUM_ALL '
                             'searches -- does not change the count); frac_r05_formula is the count '
                             'of rounds 1-5 (a forward transform and a mixing multiply per (bin, segment)) over the same time, for comparison',
                'hbm_note': 'no length-N intermediate exists on this path: HBM traffic per launch is the 8 MiB block plus partial sums '
                            '(see traffic); the two-pass algorithmic bytes below are context, not bytes moved',
                'twopass_formulation_alg_bytes_per_block': b_alg(Dl_, Mu, N),
                'twopass_alg_bytes_over_time_GBps': round(b_alg(Dl_, Mu, N) / t_block_dev / 1e9, 1)})
            return core

        if pinfo['path'] == 'segment':
            roof = segment_roofline(pinfo, Dl, counts, kms)
        else:
            roof = twopass_roofline(Dl, counts, kms, tun, timed_steps)
        roof['pipeline'] = {'device_ms_per_block': round(t_block_dev * 1e3, 4),
                            'B_alg_twopass_per_block': b_alg(Dl, Mu, N), 'B_ref_unfused_per_block': b_ref(Dl, M, N)}
        pshort = pinfo['path'] + (f' L=2^{pinfo["log2L"]}, {pinfo["taps"]} taps' if pinfo['path'] == 'segment' else '')
        # (<= 120 characters: the driver's record keeps that much of a string)
        if G == 1 and log2N == 20 and args.bins == 256 and args.protocol == 'bench_GMSK':
            workload = f'C2: 1 MI355X, D=256 Doppler bins, M=8 GMSK filters (bench_GMSK), N=2^20 complex64, ov=2^10; {pshort}'
        elif G == 1 and not args.force_dist:
            workload = f'1 MI355X, D={D_total} Doppler bins, M={M} filters of {args.protocol} ({sps_bank} samples/symbol), N=2^{log2N}; {pshort}'
        elif by_blocks:
            workload = f'block round-robin: {G} GPUs, each the full D={D_total} bank on other time blocks, M={M}, N=2^{log2N}; {pshort}'
        else:
            workload = f'C4-style: D={D_total} bins, {Dl}/GPU over {G} GPUs, M={M}, N=2^{log2N}; block bcast + 1 RCCL score exchange/block; {pshort}'
        out = {
            'metric': 'IQ Msamples/sec through Doppler matched-filter bank (256 bins, 2^20 chunk)',
            'value': round(value, 3), 'unit': 'Msamples/s', 'n_gpus': G, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': workload, 'path': pinfo,
                'D_total': D_total, 'D_per_gpu': Dl, 'M': M, 'M_unique': Mu, 'log2N': log2N, 'overlap': ov,
                'signal': ('S1: GMSK bench packet at +fs/4, tiled, AWGN 10 dB (RandomState(1)), resident in HBM' if args.signal == 'S1' else
                           'S2: unit-variance white noise (RandomState(0)), sixteen blocks resident in HBM'),
                'rangeRateMax_used': rr, 'twopass_tuning(chunk,mpb,rows,jsplit)': list(tun),
                'units': 'samples through a 256-bin bank, summed over ranks',
                'stream_msamples': round((N - ov) / (elapsed / args.steps) / 1e6, 3) if not by_blocks else round(value, 3),
                'world_size': G, 'backend': (('nccl (RCCL)' if args.backend == 'nccl' else 'gloo (rehearsal)') if dist is not None else None),
                'rccl_world': (dist.get_world_size() if dist is not None and args.backend == 'nccl' else None),
                'rank_devices': rank_devices, 'distinct_devices': len({(d['device'], d['uuid']) for d in rank_devices}),
                'broadcast_mode': (None if shard is None else ('own communicator and stream, one block ahead' if not args.single_comm
                                                               else 'single communicator, single stream')
                                   + ('' if not args.no_prefetch else ', no prefetch')),
                'carrier_found': carrier_ok if carrier_ok is None else bool(carrier_ok),
            },
            'roofline': roof,
        }
        out['config'].update(extras)
        # the same figures as flat scalars (a reader that keeps only scalar config keys still sees every single-GPU config)
        per_step = (N - ov) * G / 1e6
        flat = {
            'repeats': args.repeats, 'untimed_steps_before': first - timed_steps,
            'ms_per_step_min': round(min(times) / args.steps * 1e3, 4), 'ms_per_step_max': round(max(times) / args.steps * 1e3, 4),
            'value_min': round(per_step / (max(times) / args.steps), 3), 'value_max': round(per_step / (min(times) / args.steps), 3),
            'roofline_frac': roof.get('frac'), 'roofline_bound': roof.get('bound'), 'roofline_frac_r05_formula': roof.get('frac_r05_formula')}
        if s2 is not None:
            # the headline's loop on S2: median repeat, the same flop formula on the HIP-event time of its own launches
            el2 = float(np.median(s2['times']))
            s2_roof = segment_roofline(pinfo, Dl, s2['counts'], s2['kms']) if pinfo['path'] == 'segment' else \
                twopass_roofline(Dl, s2['counts'], s2['kms'], tun, timed_steps)
            out['config']['s2'] = {
                'signal': 'S2: unit-variance white noise (RandomState(0)), sixteen blocks resident in HBM; same loop, settle rule, '
                          '--steps and --repeats as the headline',
                'ms_per_step': round(el2 / args.steps * 1e3, 4), 'msamples': round(per_step / (el2 / args.steps), 3),
                'ms_per_step_min': round(min(s2['times']) / args.steps * 1e3, 4), 'ms_per_step_max': round(max(s2['times']) / args.steps * 1e3, 4),
                'untimed_steps_before': s2['untimed_steps_before'], 'roofline_frac': s2_roof['frac'], 'avg_launch_ms': s2_roof['avg_launch_ms'],
                'over_s1': round(elapsed / el2, 4),
                'note': 'profiles/r06_s1_vs_s2.md: same cycles per launch, same clock, same watts as S1; the -10 ... -13 % of earlier '
                        'rounds was an 8-step timing inside the clock ramp that follows host-side noise generation'}
            flat['s2_msamples'] = out['config']['s2']['msamples']
            flat['s2_roofline_frac'] = s2_roof['frac']
            flat['s2_over_s1'] = out['config']['s2']['over_s1']
        for key, fig in [(b['protocol'].replace('bench_', '').lower(), b) for b in extras.get('other_banks', [])] + \
                        ([('c3', extras['c3'])] if 'c3' in extras else []):
            flat[f'{key}_msamples'] = fig['msamples']
            flat[f'{key}_ms_per_step'] = fig['ms_per_step']
            flat[f'{key}_roofline_frac'] = fig.get('roofline', {}).get('frac')
            flat[f'{key}_frac_r05_formula'] = fig.get('roofline', {}).get('frac_r05_formula')
            if 's2_msamples' in fig:
                flat[f'{key}_s2_msamples'] = fig['s2_msamples']
                flat[f'{key}_s2_over_s1'] = fig['s2_over_s1']
            if 'msamples' in fig.get('span_basis', {}):
                flat[f'span_{key}_msamples'] = fig['span_basis']['msamples']
                flat[f'span_{key}_max_rel_diff'] = fig['span_basis']['max_rel_diff_vs_default_search']
            if 'msamples' in fig.get('twopass', {}):
                flat[f'{key}_twopass_msamples'] = fig['twopass']['msamples']
                flat[f'{key}_twopass_hbm_frac'] = fig['twopass']['frac']
        if 'sync_correlator' in extras:
            flat['sync_streams_per_s'] = extras['sync_correlator']['streams_per_s']
        for key in ('span_basis_search', 'energy_search'):
            if key in extras:
                flat[f'{key}_msamples'] = extras[key]['msamples']
        if other is not None:
            o_info, o_t, o_counts, o_kms, o_tun, osteps, o_sinfo = other
            o_roof = (segment_roofline(o_info, Dl, o_counts, o_kms, o_sinfo) if o_info['path'] == 'segment'
                      else twopass_roofline(Dl, o_counts, o_kms, o_tun, osteps))
            o_roof['ms_per_step'] = round(o_t * 1e3, 4)
            o_roof['msamples'] = round((N - ov) / o_t / 1e6, 2)
            out['roofline_other_path'] = {'path': o_info['path'], **o_roof}
            flat[f"{o_info['path']}_msamples"] = o_roof['msamples']
            flat[f"{o_info['path']}_{'hbm' if o_roof['bound'] == 'hbm' else 'valu'}_frac"] = o_roof['frac']
            if o_roof.get('traffic') and o_roof.get('alg_bytes_per_launch'):
                flat[f"{o_info['path']}_traffic_over_alg"] = round(o_roof['traffic'] / o_roof['alg_bytes_per_launch'], 3)
        if chain:
            flat.update(chain)
        if c5:
            flat.update(c5)
        if blocks_leg:
            flat.update(blocks_leg)
        if shard is not None:
            flat['dist_mode'] = ('single-comm' if args.single_comm else 'concurrent-broadcast') + ('' if not args.no_prefetch else ', no-prefetch')
            if args.fallback_from:
                flat['fallback_from'] = args.fallback_from[:110]
        flat['stream_msamples'] = out['config']['stream_msamples']
        # ORDER: `workload`, then the scalars a reader that keeps only the first twenty scalar keys must see (key names <= 40
        # characters, strings <= 120), then everything else
        LEAD = ['roofline_frac', 'dist_mode', 'fallback_from', 'blocks_stream_msamples', 'blocks_efficiency_vs_1gpu',
                's2_msamples', 's2_over_s1', 's2_roofline_frac', 'cc11xx_msamples', 'cc11xx_roofline_frac', 'cc11xx_s2_msamples',
                'bpsk_msamples', 'bpsk_roofline_frac', 'c3_msamples', 'c3_roofline_frac', 'c3_twopass_msamples', 'c3_twopass_hbm_frac',
                'c5_sum_over_alone', 'c5_cc11xx_beside', 'c5_bpsk_beside', 'recv_n15_d64_msamples', 'chain_n15_d64_msamples',
                'chain_auto_n15_d64_msamples', 'stream_msamples',
                # (behind the first twenty)
                'c5_cc11xx_alone', 'c5_bpsk_alone', 'span_cc11xx_msamples', 'span_bpsk_msamples', 'recv_auto_n15_d64_msamples',
                'recv_n17_d64_msamples', 'chain_n17_d64_msamples', 'chain_auto_n17_d64_msamples', 'twopass_msamples', 'twopass_hbm_frac',
                'twopass_traffic_over_alg', 'ms_per_step_min', 'ms_per_step_max', 'repeats', 'sync_streams_per_s']
        lead = {k: flat[k] for k in LEAD if flat.get(k) is not None}
        rest = {k: v for k, v in out['config'].items() if k != 'workload' and k not in lead}
        rest.update({k: v for k, v in flat.items() if k not in lead})
        out['config'] = {'workload': out['config']['workload'], **lead, **rest}
        # the roofline object the same way: the contract's scalars first
        out['roofline'] = {**{k: roof[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic') if k in roof},
                           **{k: v for k, v in roof.items() if k not in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}}
        if not args.no_cpu_baseline and G == 1:
            # C2 on two whole blocks when ~26 s of CPU work allow -- the last timed block and its successor
            second = (last_block + 1) % nblocks
            cb, cscores_all, single = cpu_baseline(masks, shifts[lo:hi], [host_blocks[last_block], host_blocks[second]], N, ov, Dl)
            cscores = cscores_all[0]
            # the bounded CPU sample doubles as a full-size parity spot check of the last timed block (and of the second one)
            rel = float(np.abs(gscores[:len(cscores)] - cscores).max() / cscores.max())
            rel1 = float(np.abs(gscores[:len(single)] - single).max() / single.max())
            if len(cscores_all) > 1:
                bank.upload_device(blocks.data_ptr() + second * esz)
                bank.find_carrier()
                g2 = bank.get_scores()[:, 0].astype(np.float64)
                rel = max(rel, float(np.abs(g2 - cscores_all[1]).max() / cscores_all[1].max()))
            out['config']['cpu_blocks_timed'] = cb['blocks_timed']
            cb['max_rel_diff_vs_gpu'] = rel
            cb['max_rel_diff_vs_gpu_fp64_oracle'] = rel1
            cb['parity_tolerance'] = PARITY_TOL
            out['cpu_baseline'] = cb
            if not (rel < PARITY_TOL and rel1 < PARITY_TOL):
                rc = 1
        else:
            out['cpu_baseline'] = None
        if carrier_ok is False:
            rc = 1
        print(json.dumps(out), flush=True)
    bank.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        sys.stderr.write('bench.py: parity spot check or carrier check FAILED\n')
        sys.exit(rc)
    return out


if __name__ == '__main__':
    main()
