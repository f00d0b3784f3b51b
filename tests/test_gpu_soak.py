"""Bounded soak of the receive chain (tools/soak.py): create / set_filters / stream / close cycles over three filter banks,
then a long stream on one handle -- HIP-graph replay on, two ``run_stream`` calls per runner.  Device memory, host RSS and
open descriptors must stay flat: a leaked stream, event, graph, page-locked buffer or device buffer shows as growth per cycle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(400)
def test_receive_chain_soak_is_flat():
    assert not os.environ.get('MFB_NO_GRAPH')                 # the block path replays its launches from a HIP graph
    spec = importlib.util.spec_from_file_location('soak_tool', os.path.join(ROOT, 'tools', 'soak.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    lines = []
    out = tool.soak(cycles=200, per=6, long_blocks=2000, budget_s=25.0, min_cycles=30, log=lines.append)
    report = '\n'.join(lines)
    assert out['cycles'] >= 30, report
    # margins: half a MiB of device memory, one MiB of RSS and half a descriptor per cycle over the second half of the cycles;
    # 8 MiB of device memory and no descriptor over the long stream
    assert abs(out['device_mib_per_cycle']) < 0.5 and out['rss_mib_per_cycle'] < 1.0 and out['fds_per_cycle'] < 0.5, report
    # ... and in total over that half no more than one 2 MiB granule of device memory and 10 MiB of RSS (0.03 MiB and 0.16 MiB per
    # cycle -- decoders waiting for the cycle collector with their finders -- passed the per-cycle margins for five rounds)
    assert out['device_mib_second_half'] < 2.5 and out['rss_mib_second_half'] < 10.0, report
    assert out['long_blocks'] == 2000 and out['long_packets'] > 0, report
    # (one-sided: the allocator handing an 8 MiB granule BACK during the stream is not growth)
    assert -64 < out['long_device_mib'] < 8.5 and out['long_fds'] == 0 and out['long_rss_mib'] < 64, report
    # the same long stream with 16 blocks per device call: flat too, and the same blocks and packets as the one-block loop
    assert out['batched_blocks'] == 2000 and out['batched_packets'] == out['long_packets'] and out['batched_equals_one_block_loop'], report
    assert -64 < out['batched_device_mib'] < 8.5 and out['batched_fds'] == 0 and out['batched_rss_mib'] < 64, report
    assert out['ok'], report
