"""The search step under different stimuli on ONE handle and ONE device: S1 (the reference's GMSK bench packet at +fs/4, tiled, AWGN 10 dB),
S2 (unit-variance white noise: what a receiver sees between passes), Z (all-zero samples: the same instruction stream with nothing toggling).

    python tools/s1_vs_s2.py ab  [protocol] [D] [rounds] [steps]       interleaved timing, every leg behind its own 60 ms of untimed settle
    python tools/s1_vs_s2.py one <S1|S2|Z|...> [protocol] [D] [steps]  one stimulus only (rocprofv3 --pmc / power sampling beside it)

`ab` prints one line per (round, stimulus): wall ms per step (barrier-to-barrier, as bench.py times it) and the HIP-event time of the
search launches; then a summary with the medians and S2 / S1.  Stimuli beyond the three above (what makes a block expensive):
  S1x2   S1 scaled by sqrt(2)/|mean| to S2's power (exponent bits move, mantissas do not)
  N10    S1's own noise without the packet (AWGN at S1's level: -10 dB)
  S1c    S1 without the noise (the clean packet and its zero padding)
  S2q    S2 quantised to 8 bits (an SDR's ADC: the low mantissa bits of every sample are zero)
  P<n>   white noise of period n samples (P208 at C2: every 256-point segment holds the same numbers)
"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()
import torch  # noqa: E402
from pycusdr_amd import config as cfg, signals as sg  # noqa: E402
from pycusdr_amd.mfbank import MFBank  # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'ab'
log2N, ov, NB = 20, 1 << 10, 16
N = 1 << log2N


def make_bank(name, D):
    if name == 'CC11xx':
        conf, sps, ms = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D), 128, 3
    else:
        conf, sps, ms = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D), 16, (5 if name == 'bench_BPSK' else 3)
    _, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
    M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, ms)
    bank = MFBank(log2N, D, M)
    bank.set_filters(masks)
    bank.set_shifts(shifts)
    return bank


def stimulus(kind):
    """complex64 [NB, N]"""
    step = N - ov
    if kind in ('S1', 'S1x2', 'S1c', 'N10'):
        clean = sg.s1_stream(NB, N, ov, 'GMSK', 16, 153600, snr_db=None)
        noisy = sg.s1_stream(NB, N, ov, 'GMSK', 16, 153600, snr_db=10.0, seed=1)
        s = {'S1': noisy, 'S1x2': noisy, 'S1c': clean, 'N10': (noisy - clean).astype(np.complex64)}[kind]
        if kind == 'S1x2':
            s = (s * np.float32(np.sqrt(2.0 / np.mean(np.abs(s) ** 2)))).astype(np.complex64)
        return np.stack([s[b * step: b * step + N] for b in range(NB)])
    if kind in ('S2', 'S2q'):
        s = sg.s2_noise(NB, N)
        if kind == 'S2q':
            s = (np.clip(np.round(s.view(np.float32) * 32.0), -127, 127) / 32.0).astype(np.float32).view(np.complex64)
        return s
    if kind == 'Z':
        return np.zeros((NB, N), dtype=np.complex64)
    if kind.startswith('P'):
        # white noise of period P samples: with P = V (valid outputs per segment) every segment of a block holds the same numbers,
        # so the four segments a wave carries side by side -- lanes l, l + 16, l + 32, l + 48, which pass through one ALU lane in
        # consecutive cycles -- toggle nothing between them (the power experiment of profiles/r06_fft_ops.md)
        P = int(kind[1:])
        rs = np.random.RandomState(7)
        base = (rs.standard_normal(P) + 1j * rs.standard_normal(P)).astype(np.complex64)
        row = np.tile(base, -(-N // P))[:N]
        return np.stack([row] * NB)
    raise SystemExit(f'unknown stimulus {kind}')


def to_dev(host):
    t = torch.from_numpy(host.view(np.float32).reshape(NB, 2 * N)).to('cuda:0')
    torch.cuda.synchronize()
    return t


ESZ = 8 * N


def run_steps(bank, dev, first, n):
    for i in range(first, first + n):
        bank.upload_device(dev.data_ptr() + (i % NB) * ESZ)
        res = bank.find_carrier()
    return res


def settle(bank, dev, seconds=0.06):
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < seconds:
        run_steps(bank, dev, i, 8)
        i += 8
    return i


if mode == 'ab':
    name = sys.argv[2] if len(sys.argv) > 2 else 'bench_GMSK'
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    kinds = sys.argv[6].split(',') if len(sys.argv) > 6 else ['S1', 'S2', 'Z', 'S1x2', 'N10', 'S1c', 'S2q']
    bank = make_bank(name, D)
    print('path:', bank.get_search_path(), flush=True)
    devs = {k: to_dev(stimulus(k)) for k in kinds}
    wall = {k: [] for k in kinds}
    evt = {k: [] for k in kinds}
    for r in range(rounds):
        for k in kinds:
            first = settle(bank, devs[k])
            reps_w, reps_e = [], []
            for _ in range(5):
                torch.cuda.synchronize()
                bank.profile_enable(True)
                t0 = time.perf_counter()
                run_steps(bank, devs[k], first, steps)
                torch.cuda.synchronize()
                reps_w.append((time.perf_counter() - t0) / steps * 1e3)
                counts, kms = bank.profile_read()
                bank.profile_enable(False)
                reps_e.append(kms[0] / max(counts[0], 1))
                first += steps
            wall[k].append(float(np.median(reps_w)))
            evt[k].append(float(np.median(reps_e)))
            print(f'round {r} {k:5s} wall {np.median(reps_w):.4f} ms/step ({min(reps_w):.4f} .. {max(reps_w):.4f})  '
                  f'search launches {np.median(reps_e):.4f} ms', flush=True)
    base = float(np.median(wall[kinds[0]]))
    print(f'\n{name} D={D} N=2^{log2N}: medians over {rounds} interleaved rounds of 5 x {steps} steps, 60 ms settle per leg')
    for k in kinds:
        w, e = float(np.median(wall[k])), float(np.median(evt[k]))
        print(f'  {k:5s} {w:.4f} ms/step  {(N - ov) / w / 1e3:7.1f} Msamples/s   search {e:.4f} ms   x{base / w:.3f} of {kinds[0]}')
    bank.close()
else:
    kind = sys.argv[2]
    name = sys.argv[3] if len(sys.argv) > 3 else 'bench_GMSK'
    D = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    bank = make_bank(name, D)
    dev = to_dev(stimulus(kind))
    first = settle(bank, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(bank, dev, first, steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{kind} {name} D={D}: {dt * 1e3:.4f} ms/step {(N - ov) / dt / 1e6:.1f} Msamples/s over {steps} steps', flush=True)
    bank.close()
