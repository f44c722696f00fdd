"""bench.py's host-side helpers (no GPU): per-step statistics, the sensor reader's "never another card's numbers" rule, the traffic key."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_step_stats_median_min_max():
    b = _bench()
    s = b.step_stats([120.0, 119.0, 250.0, 118.0])
    assert s['median_ms'] == 119.5 and s['min_ms'] == 118.0 and s['max_ms'] == 250.0 and s['step_ms'] == [120.0, 119.0, 250.0, 118.0]
    assert b.step_stats([3.0, 1.0, 2.0])['median_ms'] == 2.0


def test_sensors_without_a_gpu_report_nothing_rather_than_another_card(monkeypatch):
    """No HIP device here: the PCI lookup finds no card, rocm-smi is switched off -> every field None (round 4's first version read card0
    of a box with 45 cards: the clock and power of somebody else's GPU)."""
    b = _bench()
    monkeypatch.setenv('L2I_NO_ROCM_SMI', '1')
    b.gpu_sensors.__dict__.pop('card', None)
    out = b.gpu_sensors(0)
    assert out == dict(sclk_mhz=None, power_w=None, source=None)


def test_traffic_key_is_the_kernel_source_hash():
    b = _bench()
    from latent2im_amd import _lib
    assert b.source_hash() == _lib.source_hash()


def test_call_bytes_counts_every_operand_once():
    b = _bench()
    # (start, end, flop, shape, entry, family); shape = (B, cin, cout, kh, kw, stride, H, W, OH, OW, step, in_mask, in_scale, flags)
    q = (None, None, 0.0, (2, 8, 16, 3, 3, 1, 10, 12, 10, 12, 1, False, False, 'br0'), 'l2i_conv2d_f32', 'winograd4_f32')
    assert b.call_bytes(q) == 4.0 * (2 * 8 * 120 + 2 * 16 * 120 * 2)          # input + output + one residual map
    q16 = (None, None, 0.0, (2, 8, 16, 3, 3, 1, 10, 12, 10, 12, 1, False, False, '0'), 'l2i_conv2d_h8', 'conv_h8')
    assert b.call_bytes(q16) == 2.0 * (2 * 8 * 120 + 2 * 16 * 120)
