"""load_config: the reference's config format (JSON with // comments, layered through "configBase").  When the reference
tree is present (the authoring container; never the GPU box) its own bench and CC11xx configs are loaded and compared
with the dictionaries our builders produce, on every key the demodulator reads."""
import os

import pytest

from pycusdr_amd import config as cfg

REF = '/root/reference/config'


def test_comments_layers_and_overrides(tmp_path):
    (tmp_path / 'base.json').write_text('''// base
{
  "Main": {"workerId": "base", "keep": 1}, /* block
  comment */
  "GPU": {"UHF": {"blockSize": 16, "CUDA": {"device": 0, "streams": 3}}},
  "Interfaces": {"in": "tcp://*:5512" // a URL is not a comment
  }
}''')
    sub = tmp_path / 'sub'
    sub.mkdir()
    (sub / 'mid.json').write_text('{"configBase": "../base.json", "GPU": {"UHF": {"blockSize": 15}}, "Main": {"workerId": "mid"}}')
    (sub / 'top.json').write_text('{"configBase": "mid.json", // chain\n "GPU": {"UHF": {"CUDA": {"device": 3}}}, "text": "a \\"q\\" // not a comment"}')
    c = cfg.load_config(sub / 'top.json')
    assert c['Main'] == {'workerId': 'mid', 'keep': 1}
    assert c['GPU']['UHF'] == {'blockSize': 15, 'CUDA': {'device': 3, 'streams': 3}}
    assert c['Interfaces']['in'] == 'tcp://*:5512' and c['text'] == 'a "q" // not a comment'
    assert 'configBase' not in c
    (sub / 'loop.json').write_text('{"configBase": "loop.json"}')
    with pytest.raises(ValueError):
        cfg.load_config(sub / 'loop.json')


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
@pytest.mark.parametrize('name', ['bench_GMSK', 'bench_FSK', 'bench_GFSK', 'bench_BPSK'])
def test_reference_bench_configs_equal_our_builder(name):
    ref = cfg.load_config(os.path.join(REF, 'benchmark', name + '.json'))
    ours = cfg.bench_config(name, blockSize=ref['GPU']['UHF']['blockSize'])
    assert ref['Main']['protocols']['UHF'] == name == ours['Main']['protocols']['UHF']
    for k in ('PacketLen', 'RandSeed'):
        assert ref['Main'][k] == ours['Main'][k]
    assert ref['Radios']['rangeRateMax'] == ours['Radios']['rangeRateMax']
    r, o = ref['Radios']['Rx']['UHF-H'], ours['Radios']['Rx']['UHF-H']
    for k in ('CUDA_settings', 'frequency_Hz', 'frequencyOffset_Hz', 'baud', 'samplesPerSym', 'doppCarrierSteps', 'Protocol', 'radioBackend'):
        assert r[k] == o[k], k
    for gpu_set in ('UHF', 'STX'):
        for k, v in ours['GPU'][gpu_set].items():
            assert ref['GPU'][gpu_set][k] == v, (gpu_set, k)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_reference_cc11xx_config_equals_our_builder():
    ref = cfg.load_config(os.path.join(REF, 'CC11xx.json'))
    ours = cfg.cc11xx_config(blockSize=ref['GPU']['UHF']['blockSize'])
    rx_name = next(iter(ours['Radios']['Rx']))
    r = ref['Radios']['Rx'][rx_name] if rx_name in ref['Radios']['Rx'] else next(iter(ref['Radios']['Rx'].values()))
    o = ours['Radios']['Rx'][rx_name]
    for k in ('frequency_Hz', 'frequencyOffset_Hz', 'baud', 'samplesPerSym', 'radioBackend'):
        assert r[k] == o[k], k
    assert ref['Radios']['rangeRateMax'] == ours['Radios']['rangeRateMax']
    for k in ('rx_preamble', 'rx_sync_seq'):
        assert ref['Radios']['Protocol'][k] == ours['Radios']['Protocol'][k], k
