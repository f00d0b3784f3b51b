"""Randomised equality sweep: run_stream with B blocks per device call (stream stages on the device or on the host) against the
one-block loop -- block size, bins, modulation, SNR (down to where packets are lost), B, chunk size, zero stretches (skipped blocks,
irregular blocks that go through the host code and force a re-seed of the device's state), "nothing more right now" markers of a live
source at random places (batches of 0 ... B blocks in one stream), two calls per runner.
usage: python tests/tools/fuzz_batches.py [cases] [seed]"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pycusdr_amd import config as cfg, signals as sg              # noqa: E402
from pycusdr_amd.decoder import Decoder                            # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner      # noqa: E402
from pycusdr_amd.protocol import loadProtocol                      # noqa: E402


def _same(a, b):
    return bool(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True))


def one_case(rs, log=print):
    mod, pname = [('GMSK', 'bench_GMSK'), ('FSK', 'bench_FSK'), ('GFSK', 'bench_GFSK'), ('BPSK', 'bench_BPSK')][rs.randint(0, 4)]
    bs = int(rs.choice([13, 14, 15, 16]))
    D = int(rs.choice([8, 17, 32, 64]))
    B = int(rs.choice([2, 3, 5, 8, 16]))
    snr = float(rs.choice([1.0, 4.0, 8.0, 14.0, 40.0]))
    stages = bool(rs.randint(0, 4))                      # mostly on the device
    N, ov = 1 << bs, 1 << 10
    step = N - ov
    nblocks = int(rs.randint(B + 1, 4 * B + 4))
    sig = sg.s1_stream(nblocks, N, ov, mod, snr_db=snr, seed=int(rs.randint(1, 1 << 30)))[ov:].copy()
    for _ in range(int(rs.randint(0, 3))):               # zero stretches: all-zero blocks (skipped), noiseless pieces (irregular symbols)
        a = int(rs.randint(0, len(sig)))
        sig[a:a + int(rs.randint(step // 3, 3 * step))] = 0
    chunk = int(rs.choice([1000, 4096, 16384, step, 3 * step + 17]))
    p_dry = float(rs.choice([0.0, 0.0, 0.05, 0.3]))      # how often the source says "nothing more right now" between two chunks
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    confB = copy.deepcopy(conf)
    copies = [True, False, 'auto', 'read-only'][rs.randint(0, 4)]      # chunk -> window copies: all queued for the copy thread / none / read-only chunks only
    confB['GPU']['UHF'].setdefault('HIP', {}).update(blocks_per_call=B, stream_stages=stages, async_copies='auto' if copies == 'read-only' else copies)
    if copies == 'read-only':
        sig.flags.writeable = False
    p = loadProtocol(pname)(conf=conf)
    a, b = DemodulatorRunner(conf, p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')
    da, db = Decoder(conf, p), Decoder(conf, p)
    tag = f'{mod} N=2^{bs} D={D} B={B} snr={snr} stages={stages} blocks={nblocks} chunk={chunk} dry={p_dry} copies={copies}'

    def chunks(part, size):
        for i in range(0, len(part), size):
            yield part[i:i + size]
            if p_dry and rs.rand() < p_dry:
                yield None
    try:
        cut = int(rs.randint(1, nblocks)) * step
        ok = True
        for part in (sig[:cut], sig[cut:]):              # two calls per runner: the second goes on where the first stopped
            ra, pa = a.run_stream(chunks(part, 16384), decoder=da, blocks_per_call=1)
            rb, pb = b.run_stream(chunks(part, chunk), decoder=db)
            ok = ok and len(ra) == len(rb) and len(pa) == len(pb)
            for x, y in zip(ra, rb):
                ok = ok and x['count'] == y['count'] and _same(x['data'], y['data']) and _same(x['trust'], y['trust'])
                ok = ok and all(_same(x[k], y[k]) for k in ('doppler', 'doppler_std', 'SNR', 'spSymEst', 'numSyncSig'))
            ok = ok and all(_same(u.bits, v.bits) and u.frameStartIdx == v.frameStartIdx for u, v in zip(pa, pb))
        ok = ok and _same(a.demod.poswinP, b.demod.poswinP) and _same(da.bitsOverlapBuf, db.bitsOverlapBuf)
        log(f"{'ok  ' if ok else 'FAIL'} {tag}: device stage blocks {getattr(b.demod, 'stage_blocks', 0)}, delivered searches {getattr(db, 'ahead_blocks', 0)}")
        return ok
    finally:
        a.close()
        b.close()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = sum(0 if one_case(rs) else 1 for _ in range(cases))
    print('all equal' if not bad else f'{bad} DIFFERENT')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
