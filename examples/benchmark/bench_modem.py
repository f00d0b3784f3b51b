"""Bit-error-rate benchmark through the demodulator and the decoder -- in-process counterpart of the reference's
examples/benchmark/bench_modem.py (same arguments, same SNR bookkeeping and result table, bench_modem.py:150-280).

    python examples/benchmark/bench_modem.py modscheme N SNR_low SNR_high SNR_step [--block-size 15] [--doppler-bins 64] [--search energy]

For every SNR the seed-123 bench packet (10 000 bits, create_signals.py:10-26) is sent N times, each copy with fresh
white noise at SNR_r = SNR + 10 log10(bw / fs); the samples go in 2^14-sample chunks (bench_modem.py:32) through the
ring buffer, the HIP Doppler search + demodulation and the decoder; every packet found is compared with the known bits
(Packet_bench.checkPacketData).  The reference moves the samples and the packets over ZeroMQ between processes; here
both ends are in this process.  Needs a GPU (libmfbank.so)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from pycusdr_amd import config as cfg, signals as sg            # noqa: E402
from pycusdr_amd.decoder import Decoder                          # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner    # noqa: E402
from pycusdr_amd.hostcpu import quiet_blas                       # noqa: E402
from pycusdr_amd.protocol import loadProtocol                    # noqa: E402

CHUNK = 2 ** 14


def bandwidth(modulation, baud):
    """Noise bandwidths the reference's bench uses (bench_modem.py:203-209)."""
    return {'GMSK': baud / 0.7, 'BPSK': baud * 1.5, 'FSK': 2 * baud + 2 * (baud / 2), 'GFSK': 2 * baud + 2 * (baud / 2)}[modulation]


def make_stream(modulation, n_runs, snr, block_size, seed):
    """The stimulus of one SNR row: the bench packet sent ``n_runs`` times, each copy with fresh noise, and a noise-floor tail
    that pushes the last packet through the overlap buffers.  Returns (samples complex64, payload bits, bandwidth)."""
    spSym, baud = 16, 9600
    fs = spSym * baud
    sig, bit_data = sg.get_padded_packet(modulation, spSym, fs)
    bw = bandwidth(modulation, baud)
    snr_r = snr + 10 * np.log10(bw / fs)
    rng = np.random.RandomState(seed)
    N = 1 << block_size
    parts = [sg.awgn(sig, snr_r, rng=rng).astype(np.complex64) for _ in range(n_runs)]
    # push the last packet through the overlap buffers (noise floor only: an all-zero block has no Doppler pick)
    parts.append((1e-3 * (rng.standard_normal(2 * N) + 1j * rng.standard_normal(2 * N))).astype(np.complex64))
    stream = np.concatenate(parts)
    stream.flags.writeable = False       # a recording: the chunks cut from it cannot change (the batched loop copies them on its copy thread)
    return stream, bit_data, bw


def run_snr(modulation, n_runs, snr, block_size, search, seed, doppler_bins=64, pipelined=False, blocks_per_call=1, decode=True,
            stimulus=None):
    spSym, baud = 16, 9600
    pname = 'bench_' + modulation
    conf = cfg.bench_config(pname, blockSize=block_size, doppCarrierSteps=doppler_bins)
    if blocks_per_call not in (None, 'auto'):     # B consecutive blocks per device call (mfb_receive_blocks_*); 1 = the reference's
        conf['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = int(blocks_per_call)      # loop; None / 'auto': what the source has ready
    proto = loadProtocol(pname)(conf=conf)
    # the stimulus is made before the clock starts: the rate below is the receive chain's, not the noise generator's
    stream, bit_data, bw = stimulus if stimulus is not None else make_stream(modulation, n_runs, snr, block_size, seed)
    run = DemodulatorRunner(conf, proto, 'UHF-H')
    if search != 'transforms':
        run.demod.bank.set_search_mode(search)
    # the receive chain is long-lived in the reference (one Demodulator_process and one decoder process per run): what a new
    # decoder sets up once -- its device-side finder, page-locked staging -- is not part of the per-sample rate
    dec = Decoder(conf, proto)
    dec.prepare()
    t0 = time.perf_counter()
    results, packets = run.run_stream((stream[i:i + CHUNK] for i in range(0, len(stream), CHUNK)), decoder=dec if decode else None,
                                      pipelined=pipelined)
    dt = time.perf_counter() - t0
    run.close()
    errs = [p.checkPacketData() for p in packets]
    errs = [e for e in errs if e >= 0]                  # too-short packets report -0.1 (bench_base.py:168-176)
    nsamp = len(stream)
    return dict(SNR=float(snr), EBN0=float(snr + 10 * np.log10(bw / baud)), packets=len(errs), sent=n_runs,
                bitErrors=[int(e) for e in errs], BER=float(np.mean(np.array(errs) / len(bit_data))) if errs else 1.0,
                ksamples_per_s=nsamp / dt / 1e3, blocks=len(results))


def make_cc11xx_stream(n_frames, snr, block_size, seed, payload_bytes=200):
    """A stream for the reference's production protocol (config/CC11xx.json): ``n_frames`` CC11xx frames (preamble, sync word,
    whitened length | payload | CRC, as its TX framer builds them) of random payloads, 2-FSK at 128 samples per symbol on the
    148.32 kHz IF offset, gaps of random bits between them, AWGN; a noise-floor tail pushes the last frame through the overlap."""
    from pycusdr_amd.protocol.CC11xx import frame_bits
    sps, fs = 128, 7416 * 128
    rs = np.random.RandomState(seed)
    payloads = [rs.randint(0, 256, payload_bytes).astype(np.uint8) for _ in range(n_frames)]
    bits = np.concatenate([np.concatenate((rs.randint(0, 2, 64).astype(np.uint8), frame_bits(pl, preamble=(0xAA,) * 10)))
                           for pl in payloads])
    sig = sg.modulateFSK(bits, sps)
    sig = sig * np.exp(2j * np.pi * 148320 / fs * np.arange(len(sig)))
    N = 1 << block_size
    sig = sg.awgn(sig, snr, rng=rs).astype(np.complex64)
    tail = (1e-3 * (rs.standard_normal(2 * N) + 1j * rs.standard_normal(2 * N))).astype(np.complex64)
    stream = np.concatenate((sig, tail))
    stream.flags.writeable = False       # (a recording, as in make_stream)
    return stream, payloads


def run_cc11xx(n_frames, snr, block_size, doppler_bins=64, blocks_per_call=1, decode=True, stimulus=None, seed=3):
    """The receive chain on the CC11xx geometry of config/CC11xx.json (FSK-2, 128 samples per symbol -> 384-tap filters, IF offset,
    64 bins): samples in, frames out; a frame counts when its de-whitened payload equals one that was sent."""
    conf = cfg.cc11xx_config(blockSize=block_size, doppCarrierSteps=doppler_bins, samplesPerSym=128)
    if blocks_per_call not in (None, 'auto'):
        conf['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = int(blocks_per_call)
    proto = loadProtocol('CC11xx')(conf=conf)
    proto.CRC_CHECK = 'framer'        # the stimulus is a TX-framer frame: its CRC sits inside the length-counted bytes
    stream, payloads = stimulus if stimulus is not None else make_cc11xx_stream(n_frames, snr, block_size, seed)
    run = DemodulatorRunner(conf, proto, 'UHF-H')
    dec = Decoder(conf, proto)
    dec.prepare()
    t0 = time.perf_counter()
    results, packets = run.run_stream((stream[i:i + CHUNK] for i in range(0, len(stream), CHUNK)), decoder=dec if decode else None)
    dt = time.perf_counter() - t0
    run.close()
    sent = {pl.tobytes() for pl in payloads}
    good = 0
    for pk in packets:
        try:
            data, crc_err, _ = pk.getBinaryData()
        except Exception:       # noqa: BLE001 -- a false header on noise: not a frame
            continue
        good += (not crc_err) and np.asarray(data[:-2], dtype=np.uint8).tobytes() in sent
    return dict(frames=int(good), sent=len(payloads), packets=len(packets), ksamples_per_s=len(stream) / dt / 1e3, blocks=len(results))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('modulation', choices=['GMSK', 'FSK', 'BPSK', 'GFSK'])
    ap.add_argument('nRuns', type=int)
    ap.add_argument('SNR_low', type=float)
    ap.add_argument('SNR_high', type=float)
    ap.add_argument('SNR_step', type=float)
    ap.add_argument('--block-size', type=int, default=15)
    ap.add_argument('--doppler-bins', type=int, default=64)
    ap.add_argument('--pipelined', action='store_true', help='source, demodulator and decoder as three threads (the reference: three processes)')
    ap.add_argument('--search', choices=['transforms', 'energy'], default='transforms')
    ap.add_argument('--blocks-per-call', type=lambda v: None if v == 'auto' else int(v), default=None,
                    help="consecutive blocks handed to the device per call: a number (1 = the reference's one block per turn), or "
                         "'auto' (default): whatever the source has ready")
    ap.add_argument('--out', default=None, help='write the table as JSON')
    a = ap.parse_args()
    # the noise generator's np.linalg.norm wakes one BLAS worker per core; under a container's CPU quota their spinning gets the
    # whole process throttled for most of a 100 ms period (profiles/r03_ber.md: one 70-83 ms block in some rows)
    quiet_blas()
    rows = []
    for k, snr in enumerate(np.arange(a.SNR_low, a.SNR_high + a.SNR_step / 2, a.SNR_step)):
        r = run_snr(a.modulation, a.nRuns, snr, a.block_size, a.search, seed=1000 + k, doppler_bins=a.doppler_bins,
                    pipelined=a.pipelined, blocks_per_call=a.blocks_per_call)
        rows.append(r)
        print(f"SNR {r['SNR']:5.1f} dB:\tEB/N0 {r['EBN0']:.2f} dB\tpackets {r['packets']}/{r['sent']}\tavg. BER {r['BER']:.3e}"
              f"\t({r['ksamples_per_s']:.0f} ksamples/s through the chain)", flush=True)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(dict(modulation=a.modulation, nRuns=a.nRuns, search=a.search, blockSize=a.block_size, rows=rows), f, indent=1)


if __name__ == '__main__':
    main()
