"""Protocol plugins, stimulus generator and decoder templates against fixtures captured from the
importable numpy parts of the reference (tests/golden/make_golden.py, G1/G2/G3/G5)."""
import hashlib

import numpy as np
import pytest

from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.protocol.GFSK2_base import GFSK2
from pycusdr_amd.protocol.benchmark.bench_BPSK import _nrzs_lut, decodeNRZS

BCONF = cfg.bench_config()
CCONF = cfg.cc11xx_config()
CASES = [('bench_GMSK', 'bench_GMSK', BCONF, 16, 3), ('bench_FSK', 'bench_FSK', BCONF, 16, 3),
         ('bench_GFSK', 'bench_GFSK', BCONF, 16, 3), ('bench_BPSK', 'bench_BPSK', BCONF, 16, 5),
         ('CC11xx_sps16', 'CC11xx', CCONF, 16, 3), ('CC11xx_sps128', 'CC11xx', CCONF, 128, 3)]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('name,pname,conf,sps,ms', CASES)
def test_filter_bank_small_is_bit_exact(goldens, name, pname, conf, sps, ms):
    p = loadProtocol(pname)(conf=conf)
    M, bank = p.get_filter(1024, sps, ms)
    ref = goldens[f'g1/{name}/bank_n1024']
    assert M == int(goldens[f'g1/{name}/M']) == bank.shape[0]
    assert bank.dtype == np.complex64 and bank.shape == ref.shape
    assert np.array_equal(bank.view(np.uint32), ref.view(np.uint32))
    # templates recovered from the bank agree with the reference's
    t = np.fft.ifft(np.conj(p.get_filter(4096, sps, ms)[1].astype(np.complex128)), axis=1)[:, :ms * sps]
    assert np.abs(t - goldens[f'g1/{name}/templates']).max() < 1e-6


@pytest.mark.parametrize('name,pname,conf,sps,ms', CASES)
def test_filter_bank_2pow16_digest(goldens, name, pname, conf, sps, ms):
    p = loadProtocol(pname)(conf=conf)
    _, bank = p.get_filter(1 << 16, sps, ms)
    assert np.array_equal(bank[:, :64], goldens[f'g1/{name}/n16_head'])
    assert np.array_equal(bank[:, -64:], goldens[f'g1/{name}/n16_tail'])
    assert np.array_equal(bank[:, ::4099], goldens[f'g1/{name}/n16_stride'])
    assert _sha(bank) == str(goldens[f'g1/{name}/n16_sha256'])


def test_filter_bank_2pow20_digest_gmsk(goldens):
    p = loadProtocol('bench_GMSK')(conf=BCONF)
    _, bank = p.get_filter(1 << 20, 16, 3)
    assert _sha(bank) == str(goldens['g1/bench_GMSK/n20_sha256'])
    assert np.array_equal(bank[:, ::4099], goldens['g1/bench_GMSK/n20_stride'])


def test_gfsk2_base_filter(goldens):
    _, bank = GFSK2().get_filter(1024, 16, 3)
    assert np.array_equal(bank.view(np.uint32), goldens['g1/GFSK2_sps16/bank_n1024'].view(np.uint32))


@pytest.mark.parametrize('name,pname,conf,sps,ms', CASES)
def test_symbol_luts(goldens, name, pname, conf, sps, ms):
    p = loadProtocol(pname)(conf=conf)
    bitLUT, symLUT = p.get_symbolLUT2(ms)
    assert (bitLUT is None) == bool(goldens[f'g2/{name}/bitLUT_is_none'])
    if bitLUT is not None:
        assert np.array_equal(bitLUT, goldens[f'g2/{name}/bitLUT'])
    assert np.array_equal(np.asarray(symLUT), goldens[f'g2/{name}/symbolLUT'])
    assert bool(p.SUM_ALL_MASKS_PYTHON) == bool(goldens[f'g2/{name}/sum_all_masks'])


def test_bpsk_lut_rejects_other_mask_lengths():
    assert _nrzs_lut(4).shape == (8, 2, 2)
    with pytest.raises(Exception):
        _nrzs_lut(3)


@pytest.mark.parametrize('key,pname,conf', [('bench', 'bench_GMSK', BCONF), ('CC11xx', 'CC11xx', CCONF)])
def test_decoder_templates(goldens, key, pname, conf):
    p = loadProtocol(pname)(conf=conf)
    assert np.array_equal(p.get_mask(), goldens[f'g3/{key}/mask'])
    assert np.array_equal(p.get_syncFlag(), goldens[f'g3/{key}/syncFlag'])
    assert p.numOnesHeader == goldens[f'g3/{key}/numOnesHeader']
    assert p.numOnesSyncSig == goldens[f'g3/{key}/numOnesSyncSig']
    assert p.headerTol == goldens[f'g3/{key}/headerTol'] and p.syncSigTol == goldens[f'g3/{key}/syncSigTol']
    assert p.numBitsOverlap == goldens[f'g3/{key}/numBitsOverlap'] and p.packetLen == goldens[f'g3/{key}/packetLen']


def test_loadprotocol_unknown_name():
    with pytest.raises(ImportError):
        loadProtocol('no_such_protocol')


@pytest.mark.parametrize('mod', ['GMSK', 'FSK', 'GFSK', 'BPSK'])
def test_stimulus_matches_reference(goldens, mod):
    sig, bits = sg.get_padded_packet(mod, 16, 153600)
    assert np.array_equal(bits, goldens['g5/payload_bits'])
    assert len(sig) == int(goldens[f'g5/{mod}/len'])
    assert np.array_equal(sig[9990:10200], goldens[f'g5/{mod}/head'])
    assert np.array_equal(sig[-10200:-9990], goldens[f'g5/{mod}/tail'])
    assert np.array_equal(sig[::397], goldens[f'g5/{mod}/stride'])
    assert _sha(sig.astype(np.complex64)) == str(goldens[f'g5/{mod}/sha256_c64'])


def test_awgn_and_nrzs(goldens):
    with np.errstate(divide='ignore'):
        n = sg.awgn(goldens['g5/awgn/in_head'], 10.0, rng=np.random.RandomState(1))
    assert np.array_equal(n, goldens['g5/awgn/out_seed1_snr10'])
    enc = sg.encodeNRZS(goldens['g5/nrzs/in'])
    assert np.array_equal(enc, goldens['g5/nrzs/out'])
    assert np.array_equal(decodeNRZS(enc)[1:], goldens['g5/nrzs/in'][1:])
    with pytest.raises(TypeError):
        sg.get_padded_packet('QAM')


def test_cc11xx_pn9_crc_and_framing(goldens):
    from pycusdr_amd.protocol.CC11xx import pn9_bytes, crc16_cc11xx, frame_bits
    assert np.array_equal(pn9_bytes(300), goldens['g8/pn9_300'])
    assert crc16_cc11xx(b'123456789') == 0xAEE7           # CRC-16/CMS catalogue check value
    payload = np.arange(1, 21, dtype=np.uint8)
    bits = frame_bits(payload)
    assert len(bits) == (4 + 4 + 1 + 20 + 2) * 8
    assert np.array_equal(np.packbits(bits[:64]), [0xAA] * 4 + [0xD6, 0xBA, 0xD6, 0xBA])
    assert np.packbits(bits[64:72])[0] == (22 ^ 0xFF)     # whitened length byte
    p = loadProtocol('CC11xx')(conf=CCONF)
    p.CRC_CHECK = 'framer'        # opt-in: check the CRC where the TX framer puts it (default = the reference's flag, G12)
    pk = p.Packet(np.r_[bits, np.zeros(64, np.uint8)].astype(np.float64), 0, 0)
    data, err, _ = pk.getBinaryData()
    assert pk.packetLen == 22 and not err and np.array_equal(data[:-2], payload)
    bad = bits.copy()
    bad[100] ^= 1
    assert p.Packet(np.r_[bad, np.zeros(64, np.uint8)].astype(np.float64), 0, 0).getBinaryData()[1]
