"""The bench line kept under profiles/ (written by bench.py on an MI355X at this round's HEAD) carries every key the
driver's contract names, with consistent numbers.  CPU only: reads the stored JSON."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stored_bench_line_follows_the_contract():
    line = open(os.path.join(ROOT, 'profiles', 'r06_bench.json')).read().strip().splitlines()[-1]
    d = json.loads(line)
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert base['metric'].startswith(d['metric']) and d['unit'] == 'Msamples/s'      # BASELINE adds "at 1/2/4/8 GPUs"; n_gpus says which
    for k in ('value', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    # value = (N - ov) / t
    assert abs(d['value'] - (2 ** 20 - 2 ** 10) / (d['ms_per_step'] * 1e-3) / 1e6) / d['value'] < 1e-3
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] < 1
    assert r['avg_launch_ms'] <= d['ms_per_step']                     # the dominant kernel fits inside a step
    if r['bound'] == 'valu_fp32':
        assert abs(r['achieved'] - r['flops_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e12) / r['achieved'] < 1e-3
        assert r['traffic'] < 0.01 * r['twopass_formulation_alg_bytes_per_block']      # no length-N intermediate
        # round 6: the flops the kernel that ran performs -- with the Doppler shift on the filters' side a segment's forward transform
        # is shared by bins_per_forward bins and nothing is mixed -- and the count of rounds 1-5 over the same time beside it
        D, Q, M, L, V = 256, 5042, 8, 256, 208
        fft = 5 * L * 8
        assert r['filter_side_shift'] is True and r['bins_per_forward'] == 16
        assert abs(r['flops_per_launch'] - (D * Q * M * (6 * L + fft + 4 * V) + Q * (D // 16) * fft)) < 1e6
        assert abs(r['flops_per_launch_r05_formula'] - D * Q * ((6 * L + fft) + M * (6 * L + fft + 4 * V))) < 1e6
        assert r['frac'] < r['frac_r05_formula'] < 0.8 and abs(r['frac_r05_formula'] / r['frac'] - r['flops_per_launch_r05_formula'] / r['flops_per_launch']) < 2e-3
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['max_rel_diff_vs_gpu'] < c['parity_tolerance'] == 1e-5
    # the other BASELINE-named banks and C3, each with a roofline of its own; the packed sync correlator with its device time
    banks = {b['protocol']: b for b in d['config']['other_banks']}
    assert set(banks) == {'CC11xx', 'bench_BPSK'} and d['config']['c3']['D'] == 1024
    for b in list(banks.values()) + [d['config']['c3']]:
        assert b['path']['path'] == 'segment' and 0 < b['roofline']['frac'] < 1 and b['ms_per_step'] > b['roofline']['avg_launch_ms']
    # the same figures as flat scalars (what a reader that drops nested config objects still sees), and the repeat statistics
    c = d['config']
    assert c['repeats'] >= 5 and c['ms_per_step_min'] <= d['ms_per_step'] <= c['ms_per_step_max'] and c['untimed_steps_before'] >= d['warmup']
    assert c['ms_per_step_max'] / c['ms_per_step_min'] < 1.05                           # taken at the settled clock
    assert c['cc11xx_msamples'] == banks['CC11xx']['msamples'] and c['cc11xx_roofline_frac'] == banks['CC11xx']['roofline']['frac']
    assert c['bpsk_msamples'] == banks['bench_BPSK']['msamples'] and c['c3_roofline_frac'] == c['c3']['roofline']['frac']
    assert c['twopass_msamples'] > 100 and 0.4 < c['twopass_hbm_frac'] < 1 and 1.0 <= c['twopass_traffic_over_alg'] < 1.3
    assert c['sync_streams_per_s'] == c['sync_correlator']['streams_per_s'] and c['roofline_frac'] == r['frac']
    assert banks['CC11xx']['path']['log2L'] == 11 and banks['CC11xx']['roofline']['frac_r05_formula'] > 0.50      # the wave-local 2048-point kernel
    assert banks['CC11xx']['roofline']['filter_side_shift'] and banks['bench_BPSK']['roofline']['bins_per_forward'] == 8
    # ORDER: the driver's record keeps the first twenty scalar keys of `config` (names cut at 40 characters, strings at 120):
    # `workload` and the flat figures come first, nested objects only after them
    keys = list(c)
    assert keys[0] == 'workload' and len(c['workload']) <= 120
    lead = keys[:21]
    assert all(not isinstance(c[k], (dict, list)) and len(k) <= 40 for k in lead), lead
    for k in ('roofline_frac', 's2_msamples', 's2_over_s1', 's2_roofline_frac', 'cc11xx_msamples', 'cc11xx_roofline_frac', 'cc11xx_s2_msamples',
              'bpsk_msamples', 'bpsk_roofline_frac', 'c3_msamples', 'c3_roofline_frac', 'c3_twopass_msamples', 'c3_twopass_hbm_frac',
              'c5_sum_over_alone', 'c5_cc11xx_beside', 'c5_bpsk_beside', 'recv_n15_d64_msamples', 'chain_n15_d64_msamples',
              'chain_auto_n15_d64_msamples'):
        assert k in keys[:20], k
    assert 'stream_msamples' in lead
    # VERDICT r5, Next 1: the noise-only figure is the headline's to within the repeat spread -- the -10 ... -13 % of rounds 2-5 was an
    # eight-step timing inside the clock ramp (profiles/r06_s1_vs_s2.md)
    assert 0.98 < c['s2_over_s1'] < 1.02 and abs(c['s2_msamples'] - c['s2']['msamples']) < 1e-6 and c['s2']['untimed_steps_before'] >= 8
    assert 0.98 < c['cc11xx_s2_over_s1'] < 1.02
    # Next 4: BASELINE C5 as worded -- two concurrent instances on one device, as processes and as handles
    for k in ('c5_cc11xx_alone', 'c5_bpsk_alone', 'c5_cc11xx_beside', 'c5_bpsk_beside', 'c5_inproc_cc11xx_beside', 'c5_inproc_bpsk_beside'):
        assert c[k] > 0, k
    assert 0.9 < c['c5_sum_over_alone'] < 1.1 and 0.9 < c['c5_inproc_sum_over_alone'] < 1.1
    # Next 5: a plain iterator gets the configured-B figures without being configured
    assert c['recv_auto_n15_d64_msamples'] >= 0.9 * c['recv_n15_d64_msamples'] and c['chain_auto_n15_d64_best'] >= 0.9 * c['chain_n15_d64_msamples']
    # Next 6 / 7c: the opt-in span basis on the other two banks; C3's HBM-bound formulation
    assert c['span_cc11xx_msamples'] > 1.5 * c['cc11xx_msamples'] and c['span_bpsk_msamples'] > 2 * c['bpsk_msamples']
    assert c['span_cc11xx_max_rel_diff'] < 1e-6 and c['span_bpsk_max_rel_diff'] < 1e-6
    assert 30 < c['c3_twopass_msamples'] < 60 and 0.5 < c['c3_twopass_hbm_frac'] < 0.8
    assert list(r)[:6] == ['bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'] and all(len(k) <= 40 for k in r)
    # the receive chain at the reference's own block geometry (config/base.json:13,33), B blocks per device call
    assert c['recv_n15_d64_msamples'] >= 580 and c['recv_n17_d64_msamples'] >= 900
    assert c['recv_n15_d64_msamples'] > 2 * c['recv_b1_n15_d64_msamples'] and c['chain_n15_d64_packets'].split('/')[0] == c['chain_n15_d64_packets'].split('/')[1]
    # the CPU baseline timed two whole blocks; the single-thread leg eight bins
    assert d['cpu_baseline']['blocks_timed'] >= 1 and '8 of 256 bins' in d['cpu_baseline']['single_thread']['sample']
    sc = d['config']['sync_correlator']
    assert sc['exact_vs_np_convolve_stream0'] and sc['device_ms'] < sc['call_ms'] and 0 < sc['pcie_frac_of_63GBps'] < 1
